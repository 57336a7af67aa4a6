// C ABI of the library (declared in include/plonk_hip.h): context, transform plans,
// workspace, and the host- and device-pointer entry points built on the kernels in
// ntt.cuh, grand_product.cuh and constraints.cuh.
//
// This file is ONE FIELD's instantiation of the library: it is compiled once per field (-DPLK_FIELD_ID=0, 1, ...;
// field.cuh holds the table), field_syms.h renames its entry points to plkf<ID>_... and its namespace to
// plk_field<ID>, and dispatch.cpp defines the public names, forwarding by the field of the context.
#include "field_syms.h"

#include "../../include/plonk_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "constraints.cuh"
#include "field.cuh"
#include "gate_table.cuh"
#include "grand_product.cuh"
#include "msm.cuh"
#include "ntt.cuh"
#include "quotient.cuh"
#include "sigma_table.cuh"

using namespace plk;

namespace {

constexpr size_t ELEM = 32;  // bytes per field element

// Intermediate arrays of the row-range grand product: per-row terms and per-block products (grand_product.cuh).
struct GpScratch {
  uint64_t *num = nullptr, *den = nullptr;   // rows elements each
  uint64_t *bn = nullptr, *bd = nullptr;     // one element per block of GP_BLOCK_ROWS rows
};


#ifndef PLK_GP_FUSED_APPLY
#define PLK_GP_FUSED_APPLY 1
#endif
#ifndef PLK_FIXED_RADIX
#define PLK_FIXED_RADIX 1
#endif

// The library's own column buffers (the transforms' intermediate buffer, the coefficient vectors and coset planes of
// rounds 2+3 and of the extension) are laid out this many 64-bit words further apart than their length: vectors of
// 2^k elements at multiples of their own size put the same offset of every column into the same sets of an XCD's
// L2, and a launch that walks 17 columns and 51 output slots at one offset at a time then evicts its own lines -
// 128 bytes of skew per vector take a quarter of the memory-side reads and a tenth of the writes off the
// extension's first pass (tools/ntt_harness.hip) and 2 % off its time.
#ifndef PLK_PLANE_SKEW_WORDS
#define PLK_PLANE_SKEW_WORDS 16
#endif
constexpr size_t PLANE_SKEW_WORDS = PLK_PLANE_SKEW_WORDS;
static_assert(PLANE_SKEW_WORDS % 16 == 0, "vectors stay aligned to 128 bytes");

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

struct NttPlan {
  int log_n = 0;
  int npass = 1;
  int log_r[3] = {0, 0, 0};
  NttTables tab{};
  DevBuf master, hi, hi_scaled, lo, scale, full[2], prescale4;
};

// What launch_ntt does with its NTT_IN_U64 columns beyond F::from: an optional constant every such column is
// multiplied by (Montgomery form; the transforms are linear, so the outputs are the constant times the column's),
// and an optional place for the converted inputs (canonical elements, column b of the batch at
// conv_out + (b - conv_first) * n_in elements).
struct U64Opt {
  const Fe* scale = nullptr;
  uint64_t* conv_out = nullptr;
  int conv_first = 0;
  size_t conv_stride = 0;   // 64-bit words between the converted columns (0 = n_in elements)
};

}  // namespace

// Scratch of plk_dev_grand_product_noalloc, the entry point for callers that cannot allocate: a fixed array in the
// code object instead of workspace slots.  Every GPU the library runs on holds its own copy, at its own address,
// until the process ends.  In host code `scratch` is only a handle for hipGetSymbolAddress; the address the
// kernels get is looked up once per device in plk_ctx_create and kept in the context.
constexpr size_t NOALLOC_MAX_ROWS = size_t(1) << 20;
constexpr size_t NOALLOC_MAX_BLOCKS = (NOALLOC_MAX_ROWS + GP_BLOCK_ROWS - 1) / GP_BLOCK_ROWS;
constexpr size_t NOALLOC_SCRATCH_BYTES = NOALLOC_MAX_BLOCKS * ELEM;   // one block ratio per GP_BLOCK_ROWS rows: 32 KiB
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];
static_assert(NOALLOC_MAX_ROWS == PLK_NOALLOC_MAX_ROWS, "the header's limit and the array's size are one number");
static_assert(GP_SMALL_ROWS / GP_THREADS <= NOALLOC_MAX_BLOCKS, "the small-circuit path (one row per thread) keeps its block ratios in the same array");

// Copies between ordinary (pageable) host memory and the device.  hipMemcpy stages such copies through one
// thread; here COPY_WORKERS threads (the caller and a small pool that lives with the context) each copy chunks
// into page-locked slots of their own and send them on streams of their own, copy and DMA overlapping.
constexpr size_t COPY_CHUNK = size_t(4) << 20;      // bytes per slot
constexpr size_t COPY_MIN = size_t(8) << 20;        // below this a plain hipMemcpyAsync
constexpr int COPY_WORKERS = 10;                    // at most; fewer on a host with fewer hardware threads

namespace {

struct CopyLane { void* slot[2] = {nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t ev[2] = {nullptr, nullptr}; };

struct CopyPool {
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  std::function<void(int)> job;
  uint64_t generation = 0;
  int pending = 0;
  bool stop = false;
  int workers = 1;                                    // the caller included
  void start() {
    workers = std::min<int>(COPY_WORKERS, std::max<int>(2, (int)std::thread::hardware_concurrency()));
    for (int w = 1; w < workers; ++w)
      threads.emplace_back([this, w] {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(int)> j;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_start.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            j = job;
          }
          j(w);
          std::lock_guard<std::mutex> lk(m);
          if (--pending == 0) cv_done.notify_one();
        }
      });
  }
  // f(0) on the caller, f(1..) on the pool; returns when all are done
  void run(const std::function<void(int)>& f) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = f; pending = (int)threads.size(); ++generation;
    }
    cv_start.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(m);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(m); stop = true; }
    cv_start.notify_all();
    for (auto& t : threads) t.join();
    threads.clear();
  }
};

}  // namespace

struct plk_ctx {
  int field_id = PLK_FIELD_ID;   // FIRST member: dispatch.cpp reads it to pick the instantiation
  int device = 0;
  CopyLane copy_lanes[COPY_WORKERS];
  CopyPool copy_pool;
  bool copy_ready = false;
  hipStream_t stream = nullptr;  // used by the host entry points
  hipStream_t aux = nullptr;     // side stream of plk_round23: the grand-product chain runs beside the transforms
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // (calls on different streams are ordered one after another through ev_order: StreamCall below)
  // plk_round23 on host arrays: uploads, kernels and downloads overlap on streams of their own
  hipStream_t copy_up = nullptr, copy_down = nullptr;
  hipEvent_t ev_pipe[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipStream_t order_stream = nullptr;   // the stream ev_order was last recorded on (a value to compare with, nothing more)
  bool have_order = false;
  hipEvent_t ev_order = nullptr;
  std::map<std::pair<int, int>, NttPlan> plans;  // (log_n, inverse)
  DevBuf master_sh[2];                           // in-tile twiddle pairs for the precomputed-quotient product, per direction
  struct FsPlan {                                // tables of one four-step transform seen from one rank (ntt.cuh, FsArgs)
    int log_n = 0, log_n1 = 0, log_n2 = 0, la = 0, lb = 0, lA = 0, lB = 0, log_g = 0, lo_bits = 0;
    DevBuf sub_a, sub_b, block, hi, lo;
    const uint32_t *master = nullptr, *master_sh = nullptr;
  };
  std::map<std::array<long long, 4>, FsPlan> fs_plans;   // (log N, G, rank, inverse)
  std::map<std::string, DevBuf> ws;              // grow-only workspace slots
  struct PowTable { DevBuf lo, hi; size_t hi_count = 0; };
  std::map<std::array<uint64_t, 4>, PowTable> powers;   // g -> tables of g^j (coset transforms)
  uint32_t* msm_table = nullptr;                 // affine G1 points 2^(c w) * base_i at [w * msm_nbases + i] (msm.cuh), 128 bytes each
  size_t msm_nbases = 0;
  MsmWindows msm_win = {0, 0, 0};                // how the table cuts a scalar into digits
  G1Jac* msm_sums = nullptr;                     // page-locked landing place of the sums (Jacobian, ABI form)
  size_t msm_sums_cap = 0;
  size_t gp_rows = 0;                            // rows of the grand-product range held in the workspace
  GpScratch gp_sc;                               // where that range's terms and block products are
  unsigned char* noalloc_ptr = nullptr;          // this device's address of the array `scratch`
  GpStatus* status_dev = nullptr;
  GpStatus* status_host = nullptr;  // pinned
  std::string last_error;
  long long last_row = -1;
  // optional per-launch timing (plk_profile_*): one event pair per kernel launch
  int two_pass_max_log = 20;                     // transforms up to 2^this take two passes, larger ones three (measured: 2^21, 2^22 are 3-8 % slower in two)
  int first_radix_log = 0;                       // measurement knob: the first radix of two-pass plans (0 = even split)
  int fused_constraints_min_log = 16;            // rounds 2+3 of at least 2^this rows take both constraint vectors from one kernel
  bool profiling = false;
  struct ProfRec { std::string label; hipEvent_t e0, e1; double alg_bytes; double products; };
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> event_pool;
  std::string prof_report;
};

namespace {

int fail(plk_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}

// Brackets one kernel launch with HIP events on the launch stream when profiling is on.
struct ProfScope {
  plk_ctx* ctx; hipStream_t s; size_t idx = (size_t)-1;
  // alg_bytes: compulsory bytes of the launch; products: 256-bit modular products it performs (0 = not counted)
  ProfScope(plk_ctx* c, hipStream_t st, const std::string& label, double alg_bytes, double products = 0.0) : ctx(c), s(st) {
    if (!ctx->profiling) return;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (auto& e : ev) {
      if (!ctx->event_pool.empty()) { e = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
      else if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();
        if (ev[0]) ctx->event_pool.push_back(ev[0]);   // the first event goes back to the pool
        return;
      }
    }
    ctx->prof.push_back({label, ev[0], ev[1], alg_bytes, products});
    idx = ctx->prof.size() - 1;
    (void)hipEventRecord(ev[0], s);
  }
  ~ProfScope() { if (idx != (size_t)-1) (void)hipEventRecord(ctx->prof[idx].e1, s); }
};

#define HIP_TRY(ctx, expr)                                                                          \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(ctx, e_ == hipErrorOutOfMemory ? PLK_ERR_NOMEM : PLK_ERR_HIP,                     \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                               \
  } while (0)

int log2_exact(size_t n) {
  if (n == 0 || (n & (n - 1))) return -1;
  int l = 0;
  while ((size_t(1) << l) < n) ++l;
  return l;
}

int check_domain(plk_ctx* ctx, size_t n, int* log_n) {
  const int l = log2_exact(n);
  if (l < 0) return fail(ctx, PLK_ERR_NOT_POW2, "domain size " + std::to_string(n) + " is not a power of two");
  if (l > Field::TWO_ADICITY) return fail(ctx, PLK_ERR_TWO_ADICITY, "domain size exceeds the field's two-adicity");
  *log_n = l;
  return PLK_OK;
}

int ws_get(plk_ctx* ctx, const std::string& slot, size_t bytes, void** out) {
  DevBuf& b = ctx->ws[slot];
  if (b.cap < bytes) {
    if (b.p) { HIP_TRY(ctx, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    const size_t want = std::max<size_t>(bytes, 256);
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
  }
  *out = b.p;
  return PLK_OK;
}

Fe fe_from_limbs64(const uint64_t x[4]) {
  Fe r;
  for (int i = 0; i < 4; ++i) { r.l[2 * i] = (uint32_t)x[i]; r.l[2 * i + 1] = (uint32_t)(x[i] >> 32); }
  return r;
}

bool fe_is_canonical(const Fe& a) {  // a < p
  for (int i = 7; i >= 0; --i) {
    if (a.l[i] != Field::P[i]) return a.l[i] < Field::P[i];
  }
  return false;
}

// omega_N (Montgomery form) by squaring the two-adic root down, as ark-ff's get_root_of_unity
Fe root_of_unity(int log_n) {
  Fe canon;
  for (int i = 0; i < 8; ++i) canon.l[i] = Field::ROOT[i];
  Fe w = fe_mul(canon, fe_r2());
  for (int i = 0; i < Field::TWO_ADICITY - log_n; ++i) w = fe_sqr(w);
  return w;
}

int alloc_table(plk_ctx* ctx, DevBuf& buf, uint64_t count, const Fe& base, int shift, const Fe& mul, hipStream_t s) {
  HIP_TRY(ctx, hipMalloc(&buf.p, count * TW_WORDS * 4));
  buf.cap = count * TW_WORDS * 4;
  const unsigned threads = 256;
  const unsigned blocks = (unsigned)((count + threads - 1) / threads);
  hipLaunchKernelGGL(ntt_table_kernel, dim3(blocks), dim3(threads), 0, s, (uint32_t*)buf.p, count, base, shift, mul);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

void free_plan_tables(NttPlan& p) {
  for (DevBuf* b : {&p.master, &p.hi, &p.hi_scaled, &p.lo, &p.scale, &p.full[0], &p.full[1], &p.prescale4})
    if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
}

// The conversion record of raw 64-bit integers (ntt.cuh, NttU64): 2^(256+261) mod p as a plain integer - multiplying
// a raw integer x by it (mont29) gives x * 2^256, the element F::from(x) - times `scale` if one is given.
void u64_record(const Fe* scale, uint32_t rec[TW_WORDS]) {
  static const Fe c = [] { Fe v = fe_one(); for (int i = 0; i < 261; ++i) v = fe_double(v); return v; }();   // 2^256 * 2^261 mod p
  const F9 c9 = f9_from_fe(scale ? fe_mul(*scale, c) : c);       // fe_mul(s * 2^256, 2^517) = s * 2^517
  for (int i = 0; i < TW_WORDS; ++i) rec[i] = i < 9 ? c9.l[i] : 0;
}

int get_plan(plk_ctx* ctx, int log_n, bool inverse, hipStream_t s, NttPlan** out) {
  auto key = std::make_pair(log_n, inverse ? 1 : 0);
  auto it = ctx->plans.find(key);
  if (it != ctx->plans.end()) { *out = &it->second; return PLK_OK; }
  NttPlan plan;
  // a failure midway (a full table can be 1.5 GiB) must not leave the tables built so far behind
  struct PlanGuard {
    NttPlan* p;
    ~PlanGuard() { if (p) free_plan_tables(*p); }
  } guard{&plan};
  plan.log_n = log_n;
  plan.npass = log_n <= NTT_TILE_LOG ? 1 : (log_n <= ctx->two_pass_max_log ? 2 : 3);
  const int base = log_n / plan.npass, rem = log_n % plan.npass;
  for (int p = 0; p < plan.npass; ++p) plan.log_r[p] = base + (p < rem ? 1 : 0);
  if (plan.npass == 2 && ctx->first_radix_log > 0 && ctx->first_radix_log <= NTT_TILE_LOG && log_n - ctx->first_radix_log <= NTT_TILE_LOG &&
      log_n - ctx->first_radix_log >= 3 && ctx->first_radix_log >= 3) {
    plan.log_r[0] = ctx->first_radix_log;
    plan.log_r[1] = log_n - ctx->first_radix_log;
  }

  Fe w = root_of_unity(log_n);
  Fe wm = root_of_unity(NTT_MASTER_LOG);
  if (inverse) { w = fe_inverse(w); wm = fe_inverse(wm); }
  const Fe one = fe_one();
  int rc;
  if ((rc = alloc_table(ctx, plan.master, 1ull << (NTT_MASTER_LOG - 1), wm, 0, one, s))) return rc;
  const int lo_bits = log_n / 2;
  if ((rc = alloc_table(ctx, plan.lo, 1ull << lo_bits, w, 0, one, s))) return rc;
  if ((rc = alloc_table(ctx, plan.hi, 1ull << (log_n - lo_bits), w, lo_bits, one, s))) return rc;
  plan.tab.master = (const uint32_t*)plan.master.p;
  {
    // The in-tile twiddles once more, as (constant, quotient) pairs for shoup29_x2 (fr29.cuh): omega_{2^11}^e as a
    // canonical integer and floor(that * 2^261 / p).  They depend on the direction only and are built on the host
    // (a thousand long divisions), once per context.
    DevBuf& sh = ctx->master_sh[inverse ? 1 : 0];
    if (!sh.p) {
      const size_t count = size_t(1) << (NTT_MASTER_LOG - 1);
      std::vector<uint32_t> rec(count * 2 * TW_WORDS, 0);
      Fe raw_one = fe_zero(); raw_one.l[0] = 1;
      Fe cur = fe_one();
      for (size_t e = 0; e < count; ++e) {
        const Fe plain = fe_mul(cur, raw_one);               // out of Montgomery form: the integer omega^e, below p
        const F9 w9 = f9_from_fe(plain), q9 = f9_shoup_quotient_host(plain);
        for (int i = 0; i < 9; ++i) { rec[(2 * e) * TW_WORDS + i] = w9.l[i]; rec[(2 * e + 1) * TW_WORDS + i] = q9.l[i]; }
        cur = fe_mul(cur, wm);
      }
      HIP_TRY(ctx, hipMalloc(&sh.p, rec.size() * 4));
      sh.cap = rec.size() * 4;
      HIP_TRY(ctx, hipMemcpy(sh.p, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    }
    plan.tab.master_sh = (const uint32_t*)sh.p;
  }
  plan.tab.tw_lo = (const uint32_t*)plan.lo.p;
  plan.tab.tw_hi = (const uint32_t*)plan.hi.p;
  plan.tab.tw_hi_scaled = plan.tab.tw_hi;
  plan.tab.scale = nullptr;
  plan.tab.lo_bits = lo_bits;
  if (inverse) {
    const Fe n_inv = fe_inverse(fe_from_u64(1ull << log_n));
    if (plan.npass == 1) {
      if ((rc = alloc_table(ctx, plan.scale, 1, one, 0, n_inv, s))) return rc;
      plan.tab.scale = (const uint32_t*)plan.scale.p;
    } else {
      if ((rc = alloc_table(ctx, plan.hi_scaled, 1ull << (log_n - lo_bits), w, lo_bits, n_inv, s))) return rc;
      plan.tab.tw_hi_scaled = (const uint32_t*)plan.hi_scaled.p;
    }
  }
  // full inter-pass tables while they stay modest (2^25 records = 1.5 GiB); two-level beyond that
  plan.tab.tw_full[0] = plan.tab.tw_full[1] = nullptr;
  {
    int log_rest = log_n;   // log2 of the sub-problem the pass works on
    for (int pidx = 0; pidx + 1 < plan.npass; ++pidx) {
      const int lr = plan.log_r[pidx], log_inner = log_rest - lr;
      if (log_rest <= 25) {
        Fe wp = w;   // omega_{N'} = omega_N^(N / N')
        for (int i = 0; i < log_n - log_rest; ++i) wp = fe_sqr(wp);
        const uint64_t count = 1ull << log_rest;
        HIP_TRY(ctx, hipMalloc(&plan.full[pidx].p, count * TW_WORDS * 4));
        plan.full[pidx].cap = count * TW_WORDS * 4;
        const Fe mul = (inverse && pidx == 0) ? fe_inverse(fe_from_u64(1ull << log_n)) : one;
        hipLaunchKernelGGL(ntt_full_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                           (uint32_t*)plan.full[pidx].p, lr, log_inner, wp, mul);
        HIP_TRY(ctx, hipGetLastError());
        plan.tab.tw_full[pidx] = (const uint32_t*)plan.full[pidx].p;
      }
      log_rest = log_inner;
    }
  }
  // tables may be used from another stream than the one that built them
  HIP_TRY(ctx, hipStreamSynchronize(s));
  auto ins = ctx->plans.emplace(key, plan);
  guard.p = nullptr;
  *out = &ins.first->second;
  return PLK_OK;
}

// Enqueue `batch` transforms.  Columns are given either as pointer tables (stride_in/out == 0,
// batch <= NTT_MAX_BATCH) or as one base pointer plus a stride in elements.
int launch_ntt(plk_ctx* ctx, hipStream_t s, const uint64_t* const* in, const uint32_t* kinds, size_t batch,
               size_t n_in, size_t N, bool inverse, uint64_t* const* out, size_t stride_in = 0, size_t stride_out = 0,
               int cosets = 0, const char* tmp_slot = "ntt_tmp", const U64Opt* u64opt = nullptr, size_t coset_stride = 0) {
  // coset_stride: 64-bit words between a column's coset outputs (0 = N elements)
  int log_n, rc;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (n_in > N) return fail(ctx, PLK_ERR_LENGTH, "more input elements than domain points");
  if (batch == 0) return PLK_OK;
  const bool ptr_tables = stride_in == 0 || stride_out == 0;
  if (ptr_tables && batch > NTT_MAX_BATCH) return fail(ctx, PLK_ERR_ARG, "batch exceeds PLK_MAX_BATCH");
  const size_t slots = cosets ? batch * cosets : batch;
  if (slots > 65535) return fail(ctx, PLK_ERR_ARG, "batch exceeds 65535");
  NttPlan* plan;
  if ((rc = get_plan(ctx, log_n, inverse, s, &plan))) return rc;
  const uint32_t* prescale = nullptr;
  if (cosets) {
    // evaluation on the cosets omega_4N^r <omega_N>, r = 1..3, of the 4N-point domain
    if (cosets != 3 || inverse || n_in != N || log_n + 2 > Field::TWO_ADICITY) return fail(ctx, PLK_ERR_ARG, "unsupported coset request");
    if (!plan->prescale4.p) {
      const uint64_t count = 3ull << log_n;
      HIP_TRY(ctx, hipMalloc(&plan->prescale4.p, count * TW_WORDS * 4));
      plan->prescale4.cap = count * TW_WORDS * 4;
      hipLaunchKernelGGL(ntt_prescale_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s,
                         (uint32_t*)plan->prescale4.p, log_n, 3, root_of_unity(log_n + 2));
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipStreamSynchronize(s));
    }
    prescale = (const uint32_t*)plan->prescale4.p;
  }

  NttU64 u64{};
  u64_record(u64opt ? u64opt->scale : nullptr, u64.rec);
  if (u64opt && u64opt->conv_out) { u64.conv_out = u64opt->conv_out; u64.conv_stride = u64opt->conv_stride ? (uint64_t)u64opt->conv_stride : (uint64_t)n_in * 4; u64.conv_first = u64opt->conv_first; }
  NttBatch user{};   // the caller's buffers
  const size_t n_ptr_in = stride_in ? 1 : batch, n_ptr_out = stride_out ? 1 : batch;
  for (size_t b = 0; b < n_ptr_in; ++b) {
    if (!in[b] && n_in) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    user.in[b] = in[b];
    user.kind[b] = kinds ? kinds[b] : (uint32_t)NTT_IN_FIELD;
  }
  for (size_t b = 0; b < n_ptr_out; ++b) {
    if (!out[b]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    user.out[b] = out[b];
  }
  user.stride_in = stride_in * 4;
  user.stride_out = stride_out * 4;
  uint64_t* tmp = nullptr;
  if (plan->npass > 1) {
    void* p;
    if ((rc = ws_get(ctx, tmp_slot, slots * (N * ELEM + PLANE_SKEW_WORDS * 8), &p))) return rc;
    tmp = (uint64_t*)p;
  }
  const size_t lds_bytes = NTT_LDS_BYTES;
  const int P = plan->npass;
  // algorithmic bytes of the whole transform group (read the inputs once, write the outputs once),
  // attributed in equal shares to its P launches
  // (an n -> 4n extension done as three coset transforms is credited the extension's 160 n bytes)
  const double alg_share = (cosets ? 160.0 * (double)N : 32.0 * (double)n_in + 32.0 * (double)N) * (double)batch / P;
  const std::string tag = (cosets ? "/lde" + std::to_string(log_n) + "to" + std::to_string(log_n + 2)
                                  : std::string(inverse ? "/intt" : "/ntt") + std::to_string(log_n) + (n_in < N ? "lde" : "")) +
                          (std::string(tmp_slot) == "ntt_tmp" ? "" : "/side");
  // field products per element inside a tile transform of radix 2^lr (ntt.cuh, tile_transform): the first two
  // stages cost one product per four elements (none when only a quarter of the rows is non-zero), every later pair
  // of stages one per element, a single stage left over one per two elements
  auto tile_products = [&](int lr, bool quarter) {
    if (lr < 2) return 0.0;
    return (quarter ? 0.0 : 0.25) + (double)((lr - 2) / 2) + 0.5 * (double)((lr - 2) % 2);
  };
  const bool quarter_in = (n_in << 2) <= N;
  const double elems = (double)N * (double)slots;
  // F::from(u64) on load is one more product per element of such a column (first pass only)
  double u64_products = 0.0;
  if (kinds) {
    if (stride_in) u64_products = kinds[0] == NTT_IN_U64 ? (double)n_in * (double)batch : 0.0;
    else for (size_t b = 0; b < batch; ++b) if (kinds[b] == NTT_IN_U64) u64_products += (double)n_in;
  }
  auto from_user = [&](NttBatch& b) { for (int i = 0; i < NTT_MAX_BATCH; ++i) { b.in[i] = user.in[i]; b.kind[i] = user.kind[i]; } b.stride_in = user.stride_in; };
  auto to_user = [&](NttBatch& b) { for (int i = 0; i < NTT_MAX_BATCH; ++i) b.out[i] = user.out[i]; b.stride_out = user.stride_out; };
  auto from_tmp = [&](NttBatch& b) { b.in[0] = tmp; b.stride_in = (uint64_t)N * 4 + PLANE_SKEW_WORDS; b.kind[0] = NTT_IN_FIELD; };
  auto to_tmp = [&](NttBatch& b) { b.out[0] = tmp; b.stride_out = (uint64_t)N * 4 + PLANE_SKEW_WORDS; };
  // non-final passes
  int log_inner = log_n;
  for (int p = 0; p + 1 < P; ++p) {
    const int lr = plan->log_r[p];
    log_inner -= lr;
    NttPassArgs a{};
    a.tab = plan->tab;
    a.n_in = n_in;
    a.log_n = log_n;
    a.log_r = lr;
    a.log_inner = log_inner;
    a.log_c = std::min(NTT_TILE_LOG - lr, log_inner);
    a.first = p == 0;
    a.use_scaled = p == 0;
    a.pass = p;
    a.cosets = cosets;
    a.prescale = prescale;
    a.u64 = u64;
    if (p == 0) from_user(a.batch); else from_tmp(a.batch);
    to_tmp(a.batch);
    // a pass after the first works in place: its tile must pass through LDS between loads and stores,
    // which a radix below 8 would not do (the plans never produce one)
    if (p > 0 && lr <= 2) return fail(ctx, PLK_ERR_ARG, "internal: in-place pass with a radix below 8");
    const uint64_t tiles = 1ull << (log_n - lr - a.log_c);
    if (tiles * slots > 0x7fffffffull) return fail(ctx, PLK_ERR_ARG, "transform batch too large for one launch");
    a.slots = (uint32_t)slots;
    // tiles of fewer than 4 columns (runs below 128 bytes): 4 / columns neighbouring positions side by side in the order
    a.log_group = std::max(0, std::min(2 - a.log_c, log_inner - a.log_c));
    // + the inter-pass twiddle (two products through the two-level table), + the coset prescale in the first pass
    const double per_elem = tile_products(lr, p == 0 && quarter_in) + (plan->tab.tw_full[p] ? 1.0 : 2.0) + ((p == 0 && cosets) ? 1.0 : 0.0);
    ProfScope ps(ctx, s, "ntt_pass_kernel" + tag + "/p" + std::to_string(p), alg_share, per_elem * elems + (p == 0 ? u64_products : 0.0));
    // full tiles of the large radices take the instantiation with the radix fixed at compile time (steps unrolled, each
    // with its one product form: the non-final passes of the 2^20-point plans 1.5 % faster, the direct 2^24-point transform 2 %)
    const bool fixed = PLK_FIXED_RADIX && ntt_big_radix(plan->tab.master_sh, lr) && a.log_c == NTT_TILE_LOG - lr;
    if (fixed && lr == 10) hipLaunchKernelGGL((ntt_pass_kernel<true, 10>), dim3((unsigned)(tiles * slots)), dim3(NTT_THREADS), lds_bytes, s, a);
    else if (fixed && lr == 9) hipLaunchKernelGGL((ntt_pass_kernel<true, 9>), dim3((unsigned)(tiles * slots)), dim3(NTT_THREADS), lds_bytes, s, a);
    else if (fixed && lr == 8) hipLaunchKernelGGL((ntt_pass_kernel<true, 8>), dim3((unsigned)(tiles * slots)), dim3(NTT_THREADS), lds_bytes, s, a);
    else if (ntt_big_radix(plan->tab.master_sh, lr)) hipLaunchKernelGGL(ntt_pass_kernel<true>, dim3((unsigned)(tiles * slots)), dim3(NTT_THREADS), lds_bytes, s, a);
    else hipLaunchKernelGGL(ntt_pass_kernel<false>, dim3((unsigned)(tiles * slots)), dim3(NTT_THREADS), lds_bytes, s, a);
    HIP_TRY(ctx, hipGetLastError());
  }
  {
    NttFinalArgs a{};
    a.tab = plan->tab;
    a.n_in = n_in;
    a.log_n = log_n;
    a.log_r = plan->log_r[P - 1];
    a.log_r1 = P >= 2 ? plan->log_r[0] : 0;
    a.log_r2 = P >= 3 ? plan->log_r[1] : 0;
    a.log_c = std::min(NTT_TILE_LOG - a.log_r, a.log_r1);
    a.first = P == 1;
    a.cosets = cosets;
    a.prescale = prescale;
    a.coset_stride = coset_stride ? (uint64_t)coset_stride : (uint64_t)N * 4;
    a.u64 = u64;
    if (P == 1) from_user(a.batch); else from_tmp(a.batch);
    to_user(a.batch);
    const unsigned tiles = 1u << (log_n - a.log_r - a.log_c);
    const double per_elem = tile_products(a.log_r, P == 1 && quarter_in) + (plan->tab.scale ? 1.0 : 0.0) + ((P == 1 && cosets) ? 1.0 : 0.0);
    ProfScope ps(ctx, s, "ntt_final_kernel" + tag, alg_share, per_elem * elems + (P == 1 ? u64_products : 0.0));
    // (the last pass keeps the run-time radix: its fixed-radix instantiation was measured 3 % slower)
    if (ntt_big_radix(plan->tab.master_sh, a.log_r)) hipLaunchKernelGGL(ntt_final_kernel<true>, dim3(tiles, (unsigned)slots), dim3(NTT_THREADS), lds_bytes, s, a);
    else hipLaunchKernelGGL(ntt_final_kernel<false>, dim3(tiles, (unsigned)slots), dim3(NTT_THREADS), lds_bytes, s, a);
    HIP_TRY(ctx, hipGetLastError());
  }
  return PLK_OK;
}

// Interpolation of `batch` columns (n rows, kinds as in launch_ntt) and their evaluations on the
// 4n-point domain in natural order: coefficients to coef_out[b] (n elements, may be null pointers'
// table = not wanted), evaluations to eval_out[b] (4n elements).  The three cosets beyond the trace
// domain are n-point transforms; the trace-domain points are the input column itself.
int launch_extend4(plk_ctx* ctx, hipStream_t s, const uint64_t* const* cols, const uint32_t* kinds, size_t batch, size_t n,
                   uint64_t* const* coef_out, uint64_t* const* eval_out, size_t chunk, size_t chunk_stride, size_t halo) {
  int rc, log_n;
  if ((rc = check_domain(ctx, n, &log_n))) return rc;
  if (log_n < 2 || log_n + 2 > Field::TWO_ADICITY) return fail(ctx, PLK_ERR_TWO_ADICITY, "extension needs 4 <= n and 4n within the two-adicity");
  if (batch == 0) return PLK_OK;
  if (batch > NTT_MAX_BATCH) return fail(ctx, PLK_ERR_ARG, "batch exceeds PLK_MAX_BATCH");
  if (chunk == 0) { chunk = 4 * n; chunk_stride = 4 * n; halo = 0; }
  if ((4 * n) % chunk || halo > chunk || chunk_stride < chunk + halo) return fail(ctx, PLK_ERR_ARG, "bad chunk layout");
  void *cp, *pp;
  const size_t plane = n * 4 + PLANE_SKEW_WORDS;                   // 64-bit words from one n-element vector to the next
  if ((rc = ws_get(ctx, "ext_coef", batch * plane * 8, &cp))) return rc;
  if ((rc = ws_get(ctx, "ext_planes", batch * 3 * plane * 8, &pp))) return rc;
  uint64_t* coef[NTT_MAX_BATCH]; const uint64_t* coef_in[NTT_MAX_BATCH]; uint64_t* planes[NTT_MAX_BATCH];
  Interleave4Args a{};
  for (size_t b = 0; b < batch; ++b) {
    if (!cols[b] || !eval_out[b]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    coef[b] = (coef_out && coef_out[b]) ? coef_out[b] : (uint64_t*)cp + b * plane;
    coef_in[b] = coef[b];
    planes[b] = (uint64_t*)pp + b * 3 * plane;
    a.plane0[b] = cols[b];
    a.raw_index[b] = kinds && kinds[b] == NTT_IN_U64;
    a.out[b] = eval_out[b];
  }
  a.planes = (const uint64_t*)pp;
  a.plane = plane;
  a.chunk = chunk; a.chunk_stride = chunk_stride; a.halo = halo;
  if ((rc = launch_ntt(ctx, s, cols, kinds, batch, n, n, true, coef))) return rc;
  if ((rc = launch_ntt(ctx, s, coef_in, nullptr, batch, n, n, false, planes, 0, 0, 3, "ntt_tmp", nullptr, plane))) return rc;
  ProfScope ps(ctx, s, "interleave4_kernel", 2.0 * 128.0 * (double)n * (double)batch);
  hipLaunchKernelGGL(interleave4_kernel, dim3((unsigned)((4 * n + EW_THREADS - 1) / EW_THREADS), (unsigned)batch), dim3(EW_THREADS), 0, s, a,
                     (uint64_t)n);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// ---- four-step transform over G ranks: the two local phases (ntt.cuh, ntt_fs_kernel) ------------------------------

// one sub-transform of 2^m points: a single pass up to 2^7 points (the tile then still holds 16 neighbouring
// columns), two passes beyond
void fs_digits(int m, int* hi, int* lo) {
  if (m <= 7) { *hi = 0; *lo = 0; return; }
  *hi = (m + 1) / 2; *lo = m - *hi;
}

int get_fs_plan(plk_ctx* ctx, hipStream_t s, int log_n, size_t G, size_t rank, bool inverse, plk_ctx::FsPlan** out) {
  const std::array<long long, 4> key = {log_n, (long long)G, (long long)rank, inverse ? 1 : 0};
  auto it = ctx->fs_plans.find(key);
  if (it != ctx->fs_plans.end()) { *out = &it->second; return PLK_OK; }
  const int log_g = log2_exact(G);
  plk_ctx::FsPlan p;
  p.log_n = log_n; p.log_n1 = (log_n + 1) / 2; p.log_n2 = log_n - p.log_n1; p.log_g = log_g;
  if (log_g < 0 || rank >= G || log_g > p.log_n2) return fail(ctx, PLK_ERR_ARG, "ranks must be a power of two dividing both sides of the N1 x N2 view");
  fs_digits(p.log_n1, &p.la, &p.lb);
  fs_digits(p.log_n2, &p.lA, &p.lB);
  const int log_c2 = p.log_n2 - log_g;
  if (p.lB > log_c2) return fail(ctx, PLK_ERR_ARG, "too many ranks for this transform size");
  struct Guard { plk_ctx::FsPlan* p; ~Guard() { if (p) for (DevBuf* b : {&p->sub_a, &p->sub_b, &p->block, &p->hi, &p->lo}) if (b->p) (void)hipFree(b->p); } } guard{&p};
  NttPlan* small;
  int rc;
  if ((rc = get_plan(ctx, 4, inverse, s, &small))) return rc;   // any plan of this direction: the in-tile twiddle tables
  p.master = small->tab.master; p.master_sh = small->tab.master_sh;
  Fe w = root_of_unity(log_n);
  if (inverse) w = fe_inverse(w);
  const Fe one = fe_one();
  auto sub_table = [&](DevBuf& buf, int m, int hi, int lo) -> int {
    if (lo == 0) return PLK_OK;
    Fe wm = w;
    for (int i = 0; i < log_n - m; ++i) wm = fe_sqr(wm);
    const uint64_t count = 1ull << m;
    HIP_TRY(ctx, hipMalloc(&buf.p, count * TW_WORDS * 4));
    buf.cap = count * TW_WORDS * 4;
    hipLaunchKernelGGL(ntt_full_table_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (uint32_t*)buf.p, hi, lo, wm, one);
    HIP_TRY(ctx, hipGetLastError());
    return PLK_OK;
  };
  if ((rc = sub_table(p.sub_a, p.log_n1, p.la, p.lb))) return rc;
  if ((rc = sub_table(p.sub_b, p.log_n2, p.lA, p.lB))) return rc;
  const Fe mul = inverse ? fe_inverse(fe_from_u64(1ull << log_n)) : one;
  const uint64_t block_records = 1ull << (p.log_n1 + log_c2);
  if (block_records <= (1ull << 24)) {
    HIP_TRY(ctx, hipMalloc(&p.block.p, block_records * TW_WORDS * 4));
    p.block.cap = block_records * TW_WORDS * 4;
    hipLaunchKernelGGL(ntt_block_table_kernel, dim3((unsigned)((block_records + 255) / 256)), dim3(256), 0, s, (uint32_t*)p.block.p,
                       1ull << p.log_n1, log_c2, (uint64_t)rank << log_c2, w, mul);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    p.lo_bits = log_n / 2;
    if ((rc = alloc_table(ctx, p.lo, 1ull << p.lo_bits, w, 0, one, s))) return rc;
    if ((rc = alloc_table(ctx, p.hi, 1ull << (log_n - p.lo_bits), w, p.lo_bits, mul, s))) return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(s));
  auto ins = ctx->fs_plans.emplace(key, p);
  guard.p = nullptr;
  *out = &ins.first->second;
  return PLK_OK;
}

template <int MODE>
int fs_launch(plk_ctx* ctx, hipStream_t s, const FsArgs& a, uint64_t tiles, const char* label, double elems, double per_elem) {
  if (tiles == 0 || tiles > 0x7fffffffull) return fail(ctx, PLK_ERR_ARG, "four-step phase: bad grid");
  ProfScope ps(ctx, s, label, 64.0 * elems, per_elem * elems);
  hipLaunchKernelGGL(ntt_fs_kernel<MODE>, dim3((unsigned)tiles), dim3(NTT_THREADS), NTT_LDS_BYTES, s, a);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

double fs_tile_products(int lr) { return lr < 2 ? 0.0 : 0.25 + (double)((lr - 2) / 2) + 0.5 * (double)((lr - 2) % 2); }

FsArgs fs_args(const plk_ctx::FsPlan& p, size_t rank) {
  FsArgs a{};
  a.master = p.master; a.master_sh = p.master_sh;
  a.tw_block = (const uint32_t*)p.block.p; a.tw_hi = (const uint32_t*)p.hi.p; a.tw_lo = (const uint32_t*)p.lo.p; a.lo_bits = p.lo_bits;
  a.log_c2 = p.log_n2 - p.log_g; a.log_r1 = p.log_n1 - p.log_g; a.log_n2 = p.log_n2;
  a.col0 = (uint64_t)rank << a.log_c2;
  return a;
}

// phase A: recv1 = [N1][c2] -> send2 = [k1][c2]
int launch_fs_cols(plk_ctx* ctx, hipStream_t s, const uint64_t* recv1, size_t N, size_t G, size_t rank, bool inverse, uint64_t* send2) {
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (!recv1 || !send2 || recv1 == send2) return fail(ctx, PLK_ERR_ARG, "four-step phase: distinct non-null buffers expected");
  plk_ctx::FsPlan* p;
  if ((rc = get_fs_plan(ctx, s, log_n, G, rank, inverse, &p))) return rc;
  FsArgs a = fs_args(*p, rank);
  const double elems = (double)(N / G);
  const uint64_t* src = recv1;
  if (p->lb) {
    void* tmp;
    if ((rc = ws_get(ctx, "fs_tmp", (N / G) * ELEM, &tmp))) return rc;
    a.in = src; a.out = (uint64_t*)tmp; a.tw_sub = (const uint32_t*)p->sub_a.p;
    a.log_r = p->la; a.log_lo = p->lb; a.log_hi = 0; a.log_c = std::min(NTT_TILE_LOG - a.log_r, a.log_c2);
    if ((rc = fs_launch<FS_A_FIRST>(ctx, s, a, (1ull << p->lb) << (a.log_c2 - a.log_c), "ntt_fs_kernel/cols/first", elems, fs_tile_products(a.log_r) + 1.0))) return rc;
    src = (const uint64_t*)tmp;
  }
  a.in = src; a.out = send2; a.tw_sub = nullptr;
  a.log_r = p->lb ? p->lb : p->log_n1; a.log_hi = p->lb ? p->la : 0; a.log_lo = 0; a.log_c = std::min(NTT_TILE_LOG - a.log_r, a.log_c2);
  return fs_launch<FS_A_LAST>(ctx, s, a, (1ull << a.log_hi) << (a.log_c2 - a.log_c), "ntt_fs_kernel/cols/last", elems,
                              fs_tile_products(a.log_r) + (p->block.p ? 1.0 : 2.0));
}

// phase B: recv2 = [G][r1][c2] -> out (natural: send3 = [G][c2][r1]; else transposed [r1][N2])
int launch_fs_rows(plk_ctx* ctx, hipStream_t s, const uint64_t* recv2, size_t N, size_t G, size_t rank, bool inverse, bool natural, uint64_t* out) {
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (!recv2 || !out || recv2 == out) return fail(ctx, PLK_ERR_ARG, "four-step phase: distinct non-null buffers expected");
  plk_ctx::FsPlan* p;
  if ((rc = get_fs_plan(ctx, s, log_n, G, rank, inverse, &p))) return rc;
  FsArgs a = fs_args(*p, rank);
  const double elems = (double)(N / G);
  const uint64_t* src = recv2;
  if (p->lB) {
    void* tmp;
    if ((rc = ws_get(ctx, "fs_tmp", (N / G) * ELEM, &tmp))) return rc;
    a.in = src; a.out = (uint64_t*)tmp; a.tw_sub = (const uint32_t*)p->sub_b.p;
    a.log_r = p->lA; a.log_lo = p->lB; a.log_hi = 0; a.log_c = std::min(NTT_TILE_LOG - a.log_r, p->lB);
    if ((rc = fs_launch<FS_B_FIRST>(ctx, s, a, ((uint64_t)1 << a.log_r1) << (p->lB - a.log_c), "ntt_fs_kernel/rows/first", elems, fs_tile_products(a.log_r) + 1.0))) return rc;
    src = (const uint64_t*)tmp;
  }
  a.in = src; a.out = out; a.tw_sub = nullptr;
  a.log_r = p->lB ? p->lB : p->log_n2; a.log_hi = p->lB ? p->lA : 0; a.log_lo = 0;
  if (natural) {
    a.log_c = std::min(NTT_TILE_LOG - a.log_r, a.log_r1);
    return fs_launch<FS_B_LAST_NATURAL>(ctx, s, a, (1ull << a.log_hi) << (a.log_r1 - a.log_c), "ntt_fs_kernel/rows/last", elems, fs_tile_products(a.log_r));
  }
  a.log_c = std::min(NTT_TILE_LOG - a.log_r, a.log_hi);
  return fs_launch<FS_B_LAST_TRANSPOSED>(ctx, s, a, ((uint64_t)1 << a.log_r1) << (a.log_hi - a.log_c), "ntt_fs_kernel/rows/last", elems, fs_tile_products(a.log_r));
}

int status_reset(plk_ctx* ctx, hipStream_t s) {
  HIP_TRY(ctx, hipMemsetAsync(ctx->status_dev, 0xff, sizeof(GpStatus), s));
  return PLK_OK;
}

// Grand product of the rows [first, first + rows) of an n-row circuit, first half: per-row terms,
// block products and their scans, left in the context's workspace.  totals == nullptr: the range is
// the whole circuit and the inversion is done here; otherwise the range's numerator and denominator
// products are written to totals[0..1] (device) and gp_apply_carries must run before gp_finish.
int gp_local(plk_ctx* ctx, hipStream_t s, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n, size_t first,
             size_t rows, const uint64_t beta_[4], const uint64_t gamma_[4], uint64_t* totals) {
  if (!beta_ || !gamma_) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (rows == 0 || first > n || rows > n - first) return fail(ctx, PLK_ERR_ARG, "row range outside the circuit");
  const Fe beta = fe_from_limbs64(beta_), gamma = fe_from_limbs64(gamma_);
  if (!fe_is_canonical(beta) || !fe_is_canonical(gamma)) return fail(ctx, PLK_ERR_ARG, "beta/gamma not reduced mod p");
  GpCols cols{};
  for (int k = 0; k < 4; ++k) {
    if (!wires[k] || !sigma[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    cols.v[k] = wires[k];
    cols.sigma[k] = sigma[k];
  }
  const size_t nb = (rows + GP_BLOCK_ROWS - 1) / GP_BLOCK_ROWS;
  GpScratch sc;
  {   // grow-only workspace slots of the context
    void *num, *den, *bn, *bd;
    int rc;
    if ((rc = ws_get(ctx, "gp_num", rows * ELEM, &num))) return rc;
    if ((rc = ws_get(ctx, "gp_den", rows * ELEM, &den))) return rc;
    if ((rc = ws_get(ctx, "gp_bnum", nb * ELEM, &bn))) return rc;
    if ((rc = ws_get(ctx, "gp_bden", nb * ELEM, &bd))) return rc;
    sc.num = (uint64_t*)num; sc.den = (uint64_t*)den; sc.bn = (uint64_t*)bn; sc.bd = (uint64_t*)bd;
  }
  const Fe beta_r2 = fe_mul(beta, fe_r2());
  {
    ProfScope ps(ctx, s, "gp_terms_kernel", 288.0 * (double)rows);   // 4 values + 4 indices + 4 gathers per row
    hipLaunchKernelGGL(gp_terms_kernel, dim3((unsigned)nb), dim3(GP_THREADS), 0, s, cols, (uint64_t)n, (uint64_t)first, (uint64_t)rows,
                       beta, gamma, beta_r2, sc.num, sc.den, sc.bn, sc.bd, ctx->status_dev);
  }
  HIP_TRY(ctx, hipGetLastError());
  {
    ProfScope ps(ctx, s, "gp_block_scan_kernel", 0.0);
    hipLaunchKernelGGL(gp_block_scan_kernel, dim3(1), dim3(GP_SCAN_THREADS), 0, s, sc.bn, sc.bd, (uint32_t)nb, totals);
  }
  HIP_TRY(ctx, hipGetLastError());
  ctx->gp_rows = rows;
  ctx->gp_sc = sc;
  return PLK_OK;
}

// carries of the other ranges (totals_all: 2 * ranges elements, device) into this range's block scans
int gp_apply_carries(plk_ctx* ctx, hipStream_t s, const uint64_t* totals_all, size_t ranges, size_t mine) {
  if (!totals_all || mine >= ranges || ranges > 4096) return fail(ctx, PLK_ERR_ARG, "bad range table");
  const size_t nb = (ctx->gp_rows + GP_BLOCK_ROWS - 1) / GP_BLOCK_ROWS;
  ProfScope ps(ctx, s, "gp_carry_kernel", 0.0);
  hipLaunchKernelGGL(gp_carry_kernel, dim3(1), dim3(GP_THREADS), 0, s, totals_all, (uint32_t)ranges, (uint32_t)mine,
                     ctx->gp_sc.bn, ctx->gp_sc.bd, (uint32_t)nb);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// second half: z_after[i] = z[first + i + 1] for the range's rows; z0 (may be null) receives z[0] = 1
int gp_finish(plk_ctx* ctx, hipStream_t s, uint64_t* z_after, uint64_t* z0) {
  if (!z_after) return fail(ctx, PLK_ERR_ARG, "null pointer");
  const size_t rows = ctx->gp_rows;
  const size_t nb = (rows + GP_BLOCK_ROWS - 1) / GP_BLOCK_ROWS;
  const GpScratch& sc = ctx->gp_sc;
  ProfScope ps(ctx, s, "gp_finish_kernel", 32.0 * (double)(rows + 1));   // the z column
  hipLaunchKernelGGL(gp_finish_kernel, dim3((unsigned)nb), dim3(GP_THREADS), 0, s, (const uint64_t*)sc.num, (const uint64_t*)sc.den,
                     (uint64_t)rows, (const uint64_t*)sc.bn, (const uint64_t*)sc.bd, z_after, z0);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// The whole-circuit grand product, one pass over the columns (grand_product.cuh): z_out receives n + 1 elements.
// `fixed` = a caller-provided array for the block ratios (one element per block) instead of a workspace slot.
int launch_grand_product(plk_ctx* ctx, hipStream_t s, const uint64_t* const wires[4], const uint64_t* const sigma[4],
                         size_t n, const uint64_t beta_[4], const uint64_t gamma_[4], uint64_t* z_out, uint64_t* fixed = nullptr) {
  if (!z_out || !beta_ || !gamma_) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (n == 0) {  // z = [1], written by a kernel: no host memory is read and nothing waited for (valid under graph capture too)
    hipLaunchKernelGGL(fe_fill_kernel, dim3(1), dim3(64), 0, s, z_out, (uint64_t)1, fe_one());
    HIP_TRY(ctx, hipGetLastError());
    return PLK_OK;
  }
  const Fe beta = fe_from_limbs64(beta_), gamma = fe_from_limbs64(gamma_);
  if (!fe_is_canonical(beta) || !fe_is_canonical(gamma)) return fail(ctx, PLK_ERR_ARG, "beta/gamma not reduced mod p");
  GpCols cols{};
  for (int k = 0; k < 4; ++k) {
    if (!wires[k] || !sigma[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    cols.v[k] = wires[k];
    cols.sigma[k] = sigma[k];
  }
  // small circuits: one row per thread, a block's latency being what the call waits for (grand_product.cuh)
  const bool small = n <= GP_SMALL_ROWS;
  const size_t block_rows = small ? (size_t)GP_THREADS : (size_t)GP_BLOCK_ROWS;
  const size_t nb = (n + block_rows - 1) / block_rows;
  if (nb > 0xffffffffull) return fail(ctx, PLK_ERR_ARG, "too many rows");
  uint64_t* ratio = fixed;
  if (!ratio) {
    void* p;
    int rc = ws_get(ctx, "gp_ratio", nb * ELEM, &p);
    if (rc) return rc;
    ratio = (uint64_t*)p;
  }
  // per-call constants of the kernel's radix-2^261 arithmetic (grand_product.cuh, GpConsts)
  GpConsts k9;
  {
    Fe t = beta;                                      // beta R
    for (int i = 0; i < 10; ++i) t = fe_double(t);    // beta 2^266
    k9.beta_sigma = f9_from_fe(t);
    Fe c = fe_r2();                                   // 2^512
    for (int i = 0; i < 10; ++i) c = fe_double(c);    // 2^522
    k9.beta_index = f9_from_fe(fe_mul(beta, c));      // beta 2^256 2^522 / 2^256
    t = gamma;
    for (int i = 0; i < 5; ++i) t = fe_double(t);     // gamma 32 R
    k9.gamma = f9_from_fe(t);
    t = fe_one();
    for (int i = 0; i < 5; ++i) t = fe_double(t);
    k9.one = f9_from_fe(t);
  }
  {
    ProfScope ps(ctx, s, "gp_local_z_kernel", 320.0 * (double)n, 24.0 * (double)n);   // 4 values + 4 indices + 4 gathers in, z out
    if (small)
      hipLaunchKernelGGL(gp_local_z_small_kernel, dim3((unsigned)nb), dim3(GP_THREADS), 0, s, cols, (uint64_t)n, k9,
                         z_out + 4, z_out, ratio, ctx->status_dev);
    else
      hipLaunchKernelGGL(gp_local_z_kernel, dim3((unsigned)nb), dim3(GP_THREADS), 0, s, cols, (uint64_t)n, k9,
                         z_out + 4, z_out, ratio, ctx->status_dev);
  }
  HIP_TRY(ctx, hipGetLastError());
  if (nb > 1 && small) {
    static_assert(GP_SMALL_ROWS / GP_THREADS <= GP_THREADS, "gp_apply_small_kernel: one thread per earlier block");
    ProfScope ps(ctx, s, "gp_apply_small_kernel", 0.0, (double)n);
    hipLaunchKernelGGL(gp_apply_small_kernel, dim3((unsigned)(nb - 1)), dim3(GP_THREADS), 0, s, z_out + 4, (uint64_t)n, (const uint64_t*)ratio);
    HIP_TRY(ctx, hipGetLastError());
  } else if (nb > 1 && nb <= (size_t)GP_BLOCK_ROWS && PLK_GP_FUSED_APPLY) {
    ProfScope ps(ctx, s, "gp_apply_fused_kernel", 0.0, (double)n);
    hipLaunchKernelGGL(gp_apply_fused_kernel<GP_ITEMS>, dim3((unsigned)(nb - 1)), dim3(GP_THREADS), 0, s, z_out + 4, (uint64_t)n, (const uint64_t*)ratio);
    HIP_TRY(ctx, hipGetLastError());
  } else if (nb > 1) {
    {
      ProfScope ps(ctx, s, "gp_ratio_scan_kernel", 0.0);
      hipLaunchKernelGGL(gp_ratio_scan_kernel, dim3(1), dim3(GP_SCAN_THREADS), 0, s, ratio, (uint32_t)nb);
    }
    HIP_TRY(ctx, hipGetLastError());
    {
      ProfScope ps(ctx, s, "gp_apply_kernel", 0.0, (double)n);
      hipLaunchKernelGGL(gp_apply_kernel<GP_ITEMS>, dim3((unsigned)(nb - 1)), dim3(GP_THREADS), 0, s, z_out + 4, (uint64_t)n, (const uint64_t*)ratio);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  return PLK_OK;
}

int launch_gate_eval(plk_ctx* ctx, hipStream_t s, const uint64_t* const w[4], const uint64_t* const sel[9], size_t N,
                     uint64_t* out) {
  if (!out) return fail(ctx, PLK_ERR_ARG, "null output");
  if (N == 0) return PLK_OK;
  GateCols cols{};
  for (int k = 0; k < 4; ++k) { if (!w[k]) return fail(ctx, PLK_ERR_ARG, "null wire column"); cols.w[k] = w[k]; }
  for (int k = 0; k < 9; ++k) cols.s[k] = sel ? sel[k] : nullptr;
  const unsigned blocks = (unsigned)((N + EW_THREADS - 1) / EW_THREADS);
  int ncols = 5;
  for (int k = 0; k < 9; ++k) ncols += cols.s[k] != nullptr;
  {
  ProfScope ps(ctx, s, "gate_constraint_kernel", 32.0 * ncols * (double)N, 7.0 * (double)N);
  hipLaunchKernelGGL(gate_constraint_kernel, dim3(blocks), dim3(EW_THREADS), 0, s, cols, (uint64_t)N, out);
  }
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int launch_perm_eval(plk_ctx* ctx, hipStream_t s, const uint64_t* z, const uint64_t* const sg[4], const uint64_t* const w[4],
                     size_t N, const uint64_t beta_[4], const uint64_t gamma_[4], size_t z_shift, uint64_t* out,
                     size_t index0 = 0, bool halo = false) {
  if (!out || !z || !beta_ || !gamma_) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (N == 0) return PLK_OK;
  if (!halo && z_shift >= N) z_shift %= N;
  const size_t z_len = halo ? N + z_shift : N;
  const Fe beta = fe_from_limbs64(beta_), gamma = fe_from_limbs64(gamma_);
  if (!fe_is_canonical(beta) || !fe_is_canonical(gamma)) return fail(ctx, PLK_ERR_ARG, "beta/gamma not reduced mod p");
  PermCols cols{};
  cols.z = z;
  for (int k = 0; k < 4; ++k) {
    if (!w[k] || !sg[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    cols.w[k] = w[k];
    cols.s[k] = sg[k];
  }
  const Fe beta_r2 = fe_mul(beta, fe_r2());
  const unsigned blocks = (unsigned)((N + EW_THREADS - 1) / EW_THREADS);
  {
  ProfScope ps(ctx, s, "perm_constraint_kernel", 320.0 * (double)N, 18.0 * (double)N);
  hipLaunchKernelGGL(perm_constraint_kernel, dim3(blocks), dim3(EW_THREADS), 0, s, cols, (uint64_t)N, (uint64_t)index0,
                     (uint64_t)z_len, beta, gamma, beta_r2, (uint64_t)z_shift, out);
  }
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// Pieces of rounds 2+3 on the 4n-point domain seen as four cosets of the trace domain.  On coset 0 every column IS
// its table (interpolation followed by evaluation at the same points is the identity), so only cosets 1..3 are
// computed: three prescaled n-point transforms per column, kept coset-major in the workspace.
// Column numbering inside the workspace: wires 0-3, selectors 4-12, sigma 13-16, z 17.
struct R23Ws {
  uint64_t *coef = nullptr, *eval = nullptr, *z = nullptr;
  size_t n = 0, N = 0;
  size_t plane() const { return n * 4 + PLANE_SKEW_WORDS; }                  // 64-bit words from one n-element vector to the next
  uint64_t* col_coef(int c) const { return coef + (size_t)c * plane(); }
  uint64_t* col_eval(int c) const { return eval + (size_t)c * N * 4; }
  uint64_t* col_cos(int c) const { return eval + (size_t)c * 3 * plane(); }  // planes 1..3 of column c
  // coset form (N = 4n) only, behind the 18 columns' planes: beta * F::from(sigma_k[j]), the trace-domain plane of the
  // scaled sigma columns (r23_sigma_group writes it, r23_perm reads it)
  uint64_t* sigma_plane0(int k) const { return eval + ((size_t)54 + (size_t)k) * plane(); }
};

int r23_workspace(plk_ctx* ctx, size_t n, size_t N, uint64_t* z_out, R23Ws* ws) {
  constexpr int NCOL = 18;
  void *coef_p, *eval_p, *z_p;
  int rc;
  const size_t skewed = n * ELEM + PLANE_SKEW_WORDS * 8;          // one n-element vector and its skew
  if ((rc = ws_get(ctx, "r23_coef", (size_t)NCOL * skewed, &coef_p))) return rc;
  // 18 vectors of N elements (plain form) or 54 coset planes and 4 scaled sigma planes of n elements (coset form)
  if ((rc = ws_get(ctx, "r23_eval", std::max((size_t)NCOL * N * ELEM, (size_t)(3 * NCOL + 4) * skewed), &eval_p))) return rc;
  if ((rc = ws_get(ctx, "r23_z", (n + 1) * ELEM, &z_p))) return rc;
  ws->coef = (uint64_t*)coef_p; ws->eval = (uint64_t*)eval_p; ws->z = z_out ? z_out : (uint64_t*)z_p;
  ws->n = n; ws->N = N;
  return PLK_OK;
}

// interpolation of m columns and their three coset transforms, into the workspace columns col_of[b]
int r23_extend_group(plk_ctx* ctx, hipStream_t s, const R23Ws& ws, const uint64_t* const* cols, const uint32_t* kinds, const int* col_of,
                     size_t m, const char* tmp_slot = "ntt_tmp", const U64Opt* u64opt = nullptr) {
  if (m == 0) return PLK_OK;
  const uint64_t* in[NTT_MAX_BATCH]; uint64_t* out[NTT_MAX_BATCH];
  int rc;
  for (size_t b = 0; b < m; ++b) { in[b] = cols[b]; out[b] = ws.col_coef(col_of[b]); }
  if ((rc = launch_ntt(ctx, s, in, kinds, m, ws.n, ws.n, true, out, 0, 0, 0, tmp_slot, u64opt))) return rc;
  for (size_t b = 0; b < m; ++b) { in[b] = ws.col_coef(col_of[b]); out[b] = ws.col_cos(col_of[b]); }
  return launch_ntt(ctx, s, in, nullptr, m, ws.n, ws.n, false, out, 0, 0, 3, tmp_slot, nullptr, ws.plane());
}

// The sigma columns (positions first .. first + 3 of a group, raw indices): in the coset form they are transformed
// times beta - the permutation check needs beta * sigma_k(x) only, and a constant factor passes through the
// transforms - with beta * F::from(sigma_k[j]) itself kept as the trace-domain plane.
struct SigmaScale {
  Fe beta; U64Opt opt;
  SigmaScale(const R23Ws& ws, const uint64_t beta_[4], int first) : beta(fe_from_limbs64(beta_)) {
    opt.scale = &beta; opt.conv_out = ws.sigma_plane0(0); opt.conv_first = first; opt.conv_stride = ws.plane();
  }
};

// grand product -> Z interpolation -> Z cosets (column 17)
int r23_z_chain(plk_ctx* ctx, hipStream_t s, const R23Ws& ws, const uint64_t* const wires[4], const uint64_t* const sigma[4],
                const uint64_t beta[4], const uint64_t gamma[4], hipEvent_t z_ready = nullptr) {
  int rc;
  if ((rc = launch_grand_product(ctx, s, wires, sigma, ws.n, beta, gamma, ws.z))) return rc;
  if (z_ready) HIP_TRY(ctx, hipEventRecord(z_ready, s));
  const uint64_t* zin[1] = {ws.z}; const uint32_t kind[1] = {NTT_IN_FIELD}; const int col[1] = {17};
  return r23_extend_group(ctx, s, ws, zin, kind, col, 1, "ntt_tmp_aux");
}

int r23_gate(plk_ctx* ctx, hipStream_t s, const R23Ws& ws, const uint64_t* const wires[4], const uint64_t* const sel[9], uint64_t* gate_out) {
  GateColsCoset g{};
  int ncols = 5;
  for (int k = 0; k < 4; ++k) { g.w0[k] = wires[k]; g.w[k] = ws.col_cos(k); }
  for (int k = 0; k < 9; ++k) { const bool on = sel && sel[k]; g.s0[k] = on ? sel[k] : nullptr; g.s[k] = on ? ws.col_cos(4 + k) : nullptr; ncols += on; }
  ProfScope ps(ctx, s, "gate_constraint_coset_kernel", 32.0 * ncols * (double)ws.N, 7.0 * (double)ws.N);
  hipLaunchKernelGGL(gate_constraint_coset_kernel, dim3((unsigned)((ws.N + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, s, g,
                     (uint64_t)ws.n, (uint64_t)ws.plane(), gate_out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int r23_perm(plk_ctx* ctx, hipStream_t s, const R23Ws& ws, const uint64_t* const wires[4], const uint64_t* const sigma[4],
             const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* perm_out) {
  const Fe fb = fe_from_limbs64(beta), fg = fe_from_limbs64(gamma);
  PermColsCoset pc{};
  pc.z0 = ws.z; pc.z = ws.col_cos(17);
  (void)sigma;   // (its trace-domain plane comes scaled from the workspace)
  for (int k = 0; k < 4; ++k) { pc.sig0[k] = ws.sigma_plane0(k); pc.sig[k] = ws.col_cos(13 + k); pc.w0[k] = wires[k]; pc.w[k] = ws.col_cos(k); }
  ProfScope ps(ctx, s, "perm_constraint_coset_kernel", 320.0 * (double)ws.N, 9.0 * (double)ws.N);
  hipLaunchKernelGGL(perm_constraint_coset_kernel, dim3((unsigned)((ws.N + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, s, pc,
                     (uint64_t)ws.n, (uint64_t)ws.plane(), fg, fe_mul(fb, fe_r2()), (uint64_t)(z_shift % ws.N), perm_out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// both vectors from one pass over the planes (PLK_FUSED_CONSTRAINTS=0 keeps the two kernels)
#ifndef PLK_FUSED_CONSTRAINTS
#define PLK_FUSED_CONSTRAINTS 1
#endif
int r23_gate_and_perm(plk_ctx* ctx, hipStream_t s, const R23Ws& ws, const uint64_t* const wires[4], const uint64_t* const sel[9],
                      const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* gate_out, uint64_t* perm_out) {
  const Fe fb = fe_from_limbs64(beta), fg = fe_from_limbs64(gamma);
  GateColsCoset g{};
  PermColsCoset pc{};
  int ncols = 5 + 10;       // wires, gate vector; sigma, Z twice, permutation vector
  for (int k = 0; k < 4; ++k) { g.w0[k] = wires[k]; g.w[k] = ws.col_cos(k); pc.w0[k] = wires[k]; pc.w[k] = ws.col_cos(k); }
  for (int k = 0; k < 9; ++k) { const bool on = sel && sel[k]; g.s0[k] = on ? sel[k] : nullptr; g.s[k] = on ? ws.col_cos(4 + k) : nullptr; ncols += on; }
  pc.z0 = ws.z; pc.z = ws.col_cos(17);
  for (int k = 0; k < 4; ++k) { pc.sig0[k] = ws.sigma_plane0(k); pc.sig[k] = ws.col_cos(13 + k); }
  ProfScope ps(ctx, s, "round23_constraints_coset_kernel", 32.0 * ncols * (double)ws.N, 16.0 * (double)ws.N);
  hipLaunchKernelGGL(round23_constraints_coset_kernel, dim3((unsigned)((ws.N + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, s, g, pc,
                     (uint64_t)ws.n, (uint64_t)ws.plane(), fg, fe_mul(fb, fe_r2()), (uint64_t)(z_shift % ws.N), gate_out, perm_out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

bool r23_coset_form(size_t n, size_t N, int log_n) { return N == 4 * n && log_n >= 2 && log_n + 2 <= Field::TWO_ADICITY; }

// Makes `s` wait for the side stream before the enclosing function returns, whatever happens in between.
struct SideJoin {
  plk_ctx* ctx; hipStream_t s, sa; bool recorded = false, joined = false;
  void record() { if (!recorded) { (void)hipEventRecord(ctx->ev_join, sa); recorded = true; } }
  void join() { record(); if (!joined) { (void)hipStreamWaitEvent(s, ctx->ev_join, 0); joined = true; } }
  ~SideJoin() { join(); }
};

// Rounds 2+3 on device columns.  Workspace: 18 coefficient vectors (n) and 18 evaluation
// vectors (N).  Column order inside the workspace: wires 0-3, selectors 4-12, sigma 13-16, z 17.
int launch_round23(plk_ctx* ctx, hipStream_t s, const uint64_t* const wires[4], const uint64_t* const sel[9],
                   const uint64_t* const sigma[4], size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4],
                   size_t z_shift, uint64_t* z_out, uint64_t* gate_out, uint64_t* perm_out) {
  int rc, log_n, log_N;
  if ((rc = check_domain(ctx, n, &log_n))) return rc;
  if ((rc = check_domain(ctx, N, &log_N))) return rc;
  if (n > N) return fail(ctx, PLK_ERR_LENGTH, "evaluation domain smaller than the trace");
  constexpr int NCOL = 18;
  R23Ws ws;
  if ((rc = r23_workspace(ctx, n, N, z_out, &ws))) return rc;
  uint64_t* z = ws.z;

  const uint64_t* in[NCOL]; uint64_t* out[NCOL]; uint32_t kinds[NCOL]; int col_of[NCOL]; size_t m = 0;
  auto col_coef = [&](int c) { return ws.col_coef(c); };
  auto col_eval = [&](int c) { return ws.col_eval(c); };

  if (r23_coset_form(n, N, log_n)) {
    // Side stream: grand product -> Z interpolation -> Z cosets.  Its kernels are gather-, latency-
    // and launch-bound; they run beside the 17 other columns' transforms, which are VALU-bound.
    hipStream_t sa = ctx->aux;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));
    HIP_TRY(ctx, hipStreamWaitEvent(sa, ctx->ev_fork, 0));
    SideJoin side{ctx, s, sa};
    if ((rc = r23_z_chain(ctx, sa, ws, wires, sigma, beta, gamma))) return rc;
    side.record();
    for (int k = 0; k < 4; ++k) { in[m] = wires[k]; kinds[m] = NTT_IN_FIELD; col_of[m] = k; ++m; }
    for (int k = 0; k < 9; ++k) if (sel && sel[k]) { in[m] = sel[k]; kinds[m] = NTT_IN_FIELD; col_of[m] = 4 + k; ++m; }
    const SigmaScale sg_scale(ws, beta, (int)m);
    for (int k = 0; k < 4; ++k) { in[m] = sigma[k]; kinds[m] = NTT_IN_U64; col_of[m] = 13 + k; ++m; }
    if ((rc = r23_extend_group(ctx, s, ws, in, kinds, col_of, m, "ntt_tmp", &sg_scale.opt))) return rc;
    // (small circuits keep the two kernels: their critical path is the side stream's chain, and the gate vector is
    // computed beside it instead of behind it - 0.203 against 0.213 ms per call at 2^12 rows, 0.304 against 0.316 at 2^15;
    // from 2^16 rows on the one kernel wins: 0.464 against 0.469, 0.873 against 0.882 at 2^17)
    if (PLK_FUSED_CONSTRAINTS && gate_out && perm_out && log_n >= ctx->fused_constraints_min_log) {
      side.join();
      return r23_gate_and_perm(ctx, s, ws, wires, sel, beta, gamma, z_shift, gate_out, perm_out);
    }
    if (gate_out && (rc = r23_gate(ctx, s, ws, wires, sel, gate_out))) return rc;
    side.join();
    if (perm_out && (rc = r23_perm(ctx, s, ws, wires, sigma, beta, gamma, z_shift, perm_out))) return rc;
    return PLK_OK;
  }

  if ((rc = launch_grand_product(ctx, s, wires, sigma, n, beta, gamma, z))) return rc;

  // interpolation of every column present (rows 3 and 6 of the path): one batched inverse transform
  for (int k = 0; k < 4; ++k) { in[m] = wires[k]; kinds[m] = NTT_IN_FIELD; col_of[m] = k; ++m; }
  for (int k = 0; k < 9; ++k) if (sel && sel[k]) { in[m] = sel[k]; kinds[m] = NTT_IN_FIELD; col_of[m] = 4 + k; ++m; }
  for (int k = 0; k < 4; ++k) { in[m] = sigma[k]; kinds[m] = NTT_IN_U64; col_of[m] = 13 + k; ++m; }
  in[m] = z; kinds[m] = NTT_IN_FIELD; col_of[m] = 17; ++m;
  for (size_t b = 0; b < m; ++b) out[b] = col_coef(col_of[b]);
  if ((rc = launch_ntt(ctx, s, in, kinds, m, n, n, true, out))) return rc;

  // evaluation of every coefficient vector on the N-point domain (wires once, shared by both checks)
  for (size_t b = 0; b < m; ++b) { in[b] = col_coef(col_of[b]); out[b] = col_eval(col_of[b]); }
  if ((rc = launch_ntt(ctx, s, in, nullptr, m, n, N, false, out))) return rc;

  const uint64_t* w_ev[4] = {col_eval(0), col_eval(1), col_eval(2), col_eval(3)};
  const uint64_t* s_ev[9];
  for (int k = 0; k < 9; ++k) s_ev[k] = (sel && sel[k]) ? col_eval(4 + k) : nullptr;
  const uint64_t* g_ev[4] = {col_eval(13), col_eval(14), col_eval(15), col_eval(16)};
  if (gate_out && (rc = launch_gate_eval(ctx, s, w_ev, s_ev, N, gate_out))) return rc;
  if (perm_out && (rc = launch_perm_eval(ctx, s, col_eval(17), g_ev, w_ev, N, beta, gamma, z_shift, perm_out))) return rc;
  return PLK_OK;
}

// gate evaluation table of n rows (device pointers), enqueued on s
int launch_gate_table_fill(plk_ctx* ctx, hipStream_t s, const uint8_t* row_ops, const uint64_t* const node_ids[4], size_t n,
                           const uint64_t* node_values, size_t m, const uint64_t default_val[4], uint64_t* const wires_out[4],
                           uint64_t* const sel_out[9]) {
  if (n == 0) return PLK_OK;
  if (!row_ops || !default_val || (m && !node_values)) return fail(ctx, PLK_ERR_ARG, "null pointer");
  GateTableCols cols{};
  int ncols = 4;
  for (int k = 0; k < 4; ++k) {
    if (!node_ids[k] || !wires_out[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    cols.node[k] = node_ids[k];
    cols.wire[k] = wires_out[k];
  }
  for (int k = 0; k < 9; ++k) { cols.sel[k] = sel_out ? sel_out[k] : nullptr; ncols += cols.sel[k] != nullptr; }
  const Fe dv = fe_from_limbs64(default_val);
  if (!fe_is_canonical(dv)) return fail(ctx, PLK_ERR_ARG, "default_val is not a reduced field element");
  // per row: 1 op byte + 4 node ids + up to 4 gathered values in, the table's columns out
  ProfScope ps(ctx, s, "gate_table_fill_kernel", (1.0 + 32.0 + 128.0 + 32.0 * ncols) * (double)n);
  hipLaunchKernelGGL(gate_table_fill_kernel, dim3((unsigned)((n + GT_THREADS - 1) / GT_THREADS)), dim3(GT_THREADS), 0, s, cols, row_ops,
                     (uint64_t)n, node_values, (uint64_t)m, dv, fe_one());
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// tables for g^j, j < count: lo[i] = g^i (i < 1024), hi[i] = g^(1024 i)
int get_powers(plk_ctx* ctx, hipStream_t s, const Fe& g, size_t count, const uint64_t** lo, const uint64_t** hi) {
  std::array<uint64_t, 4> key;
  for (int i = 0; i < 4; ++i) key[i] = (uint64_t)g.l[2 * i] | ((uint64_t)g.l[2 * i + 1] << 32);
  plk_ctx::PowTable& t = ctx->powers[key];
  const size_t hi_count = std::max<size_t>((count + (1u << QT_LO_BITS) - 1) >> QT_LO_BITS, 1);
  if (!t.lo.p) {
    HIP_TRY(ctx, hipMalloc(&t.lo.p, (size_t)(1u << QT_LO_BITS) * ELEM));
    hipLaunchKernelGGL(pow_table_kernel, dim3((1u << QT_LO_BITS) / 256), dim3(256), 0, s, (uint64_t*)t.lo.p, (uint64_t)(1u << QT_LO_BITS), g, 0);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (t.hi_count < hi_count) {
    HIP_TRY(ctx, hipStreamSynchronize(s));                 // an older table may still be in use
    if (t.hi.p) HIP_TRY(ctx, hipFree(t.hi.p));
    t.hi.p = nullptr; t.hi_count = 0;
    HIP_TRY(ctx, hipMalloc(&t.hi.p, hi_count * ELEM));
    hipLaunchKernelGGL(pow_table_kernel, dim3((unsigned)((hi_count + 255) / 256)), dim3(256), 0, s, (uint64_t*)t.hi.p, (uint64_t)hi_count, g,
                       QT_LO_BITS);
    HIP_TRY(ctx, hipGetLastError());
    t.hi_count = hi_count;
    HIP_TRY(ctx, hipStreamSynchronize(s));                 // tables may be used from another stream later
  }
  *lo = (const uint64_t*)t.lo.p;
  *hi = (const uint64_t*)t.hi.p;
  return PLK_OK;
}

// ark-poly's coset_fft (inverse = false: n_in coefficients -> N evaluations on shift * <omega_N>) and
// coset_ifft (inverse = true: N evaluations -> N coefficients), `batch` columns
int launch_coset_ntt(plk_ctx* ctx, hipStream_t s, const uint64_t* const* in, size_t batch, size_t n_in, size_t N, const uint64_t shift_[4],
                     bool inverse, uint64_t* const* out) {
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (!shift_) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (inverse ? n_in != N : n_in > N) return fail(ctx, PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
  if (batch == 0) return PLK_OK;
  if (batch > NTT_MAX_BATCH) return fail(ctx, PLK_ERR_ARG, "batch exceeds PLK_MAX_BATCH");
  const Fe shift = fe_from_limbs64(shift_);
  if (!fe_is_canonical(shift) || fe_is_zero(shift)) return fail(ctx, PLK_ERR_ARG, "coset shift must be a non-zero reduced element");
  const uint64_t *lo, *hi;
  if ((rc = get_powers(ctx, s, inverse ? fe_inverse(shift) : shift, inverse ? N : n_in, &lo, &hi))) return rc;
  for (size_t b = 0; b < batch; ++b) if (!in[b] || !out[b]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
  if (inverse) {
    if ((rc = launch_ntt(ctx, s, in, nullptr, batch, N, N, true, out))) return rc;
    for (size_t b = 0; b < batch; ++b) {
      ProfScope ps(ctx, s, "scale_powers_kernel", 64.0 * (double)N);
      hipLaunchKernelGGL(scale_powers_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const uint64_t*)out[b], (uint64_t)N, lo, hi, out[b]);
    }
    HIP_TRY(ctx, hipGetLastError());
    return PLK_OK;
  }
  void* tp;
  if ((rc = ws_get(ctx, "coset_in", batch * std::max<size_t>(n_in, 1) * ELEM, &tp))) return rc;
  const uint64_t* scaled[NTT_MAX_BATCH];
  for (size_t b = 0; b < batch; ++b) {
    uint64_t* dst = (uint64_t*)tp + b * n_in * 4;
    scaled[b] = dst;
    if (n_in) {
      ProfScope ps(ctx, s, "scale_powers_kernel", 64.0 * (double)n_in);
      hipLaunchKernelGGL(scale_powers_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, s, in[b], (uint64_t)n_in, lo, hi, dst);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  return launch_ntt(ctx, s, scaled, nullptr, batch, n_in, N, false, out);
}

// coefficients (N) of (sum_i alpha^i nums[i]) / Z_H, nums being evaluation vectors on shift * <omega_N>
int launch_quotient(plk_ctx* ctx, hipStream_t s, const uint64_t* const* nums, size_t count, const uint64_t alpha_[4], size_t n, size_t N,
                    const uint64_t shift_[4], uint64_t* out) {
  int rc, log_n, log_N;
  if ((rc = check_domain(ctx, n, &log_n))) return rc;
  if ((rc = check_domain(ctx, N, &log_N))) return rc;
  if (n > N) return fail(ctx, PLK_ERR_LENGTH, "evaluation domain smaller than the trace domain");
  if (!nums || !alpha_ || !shift_ || !out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (count == 0 || count > QT_MAX_NUMERATORS) return fail(ctx, PLK_ERR_ARG, "between 1 and 8 numerators");
  if ((N >> log_n) > QT_MAX_RATIO) return fail(ctx, PLK_ERR_ARG, "evaluation domain more than 16 times the trace domain");
  QuotientArgs a{};
  a.count = (int)count;
  a.ratio = (uint32_t)(N >> log_n);
  a.alpha = fe_from_limbs64(alpha_);
  const Fe shift = fe_from_limbs64(shift_);
  if (!fe_is_canonical(a.alpha) || !fe_is_canonical(shift) || fe_is_zero(shift)) return fail(ctx, PLK_ERR_ARG, "alpha / shift not reduced mod p");
  for (size_t i = 0; i < count; ++i) { if (!nums[i]) return fail(ctx, PLK_ERR_ARG, "null column pointer"); a.num[i] = nums[i]; }
  // Z_H(shift * omega_N^j) = shift^n * omega_ratio^j - 1
  const Fe gn = fe_pow_u64(shift, (uint64_t)n), w = root_of_unity(log_N - log_n);
  Fe wi = fe_one();
  for (uint32_t i = 0; i < QT_MAX_RATIO; ++i) {
    if (i < a.ratio) {
      const Fe zh = fe_sub(fe_mul(gn, wi), fe_one());
      if (fe_is_zero(zh)) {
        ctx->last_row = (long long)i;
        return fail(ctx, PLK_ERR_ZERO_DENOMINATOR, "the vanishing polynomial is zero at point " + std::to_string(i) + " of this coset");
      }
      a.zinv[i] = fe_inverse(zh);
      wi = fe_mul(wi, w);
    } else {
      a.zinv[i] = fe_zero();
    }
  }
  void* tp;
  if ((rc = ws_get(ctx, "quotient_evals", N * ELEM, &tp))) return rc;
  {
    ProfScope ps(ctx, s, "quotient_combine_kernel", 32.0 * (double)(count + 1) * (double)N);
    hipLaunchKernelGGL(quotient_combine_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, a, (uint64_t)N, (uint64_t*)tp);
    HIP_TRY(ctx, hipGetLastError());
  }
  const uint64_t* in[1] = {(const uint64_t*)tp};
  uint64_t* o[1] = {out};
  return launch_coset_ntt(ctx, s, in, 1, N, N, shift_, true, o);
}

// sigma table of n rows from node ids (device pointers), enqueued on s
// Workspace and launches of the stable radix sort of (uint64 key, uint32 value) pairs (sigma_table.cuh).
// The caller fills pairs.keys[0] / vals[0] and ORs every key into *or_bits (zeroed by pair_sort_alloc);
// after pair_sort_run the sorted pairs are in buffer sig_buffer_before(or_bits, 8).
struct PairSort {
  SigPairs pairs;
  uint64_t* or_bits;
  uint32_t *hist, *sums;
  uint32_t waves, scan_tiles;
  uint64_t entries, total;
};

int pair_sort_alloc(plk_ctx* ctx, hipStream_t s, const std::string& prefix, uint64_t total, PairSort* ps, bool with_vals = true) {
  ps->total = total;
  ps->waves = (uint32_t)((total + SIG_CHUNK - 1) / SIG_CHUNK);
  ps->entries = (uint64_t)SIG_BINS * ps->waves;
  ps->scan_tiles = (uint32_t)((ps->entries + SIG_SCAN_TILE - 1) / SIG_SCAN_TILE);
  int rc;
  void *pk, *pv, *ph;
  if ((rc = ws_get(ctx, prefix + "_keys", 2 * total * 8, &pk))) return rc;
  pv = nullptr;
  if (with_vals && (rc = ws_get(ctx, prefix + "_vals", 2 * total * 4, &pv))) return rc;
  if ((rc = ws_get(ctx, prefix + "_hist", 8 + (ps->entries + ps->scan_tiles) * 4, &ph))) return rc;
  ps->pairs.keys[0] = (uint64_t*)pk; ps->pairs.keys[1] = (uint64_t*)pk + total;
  ps->pairs.vals[0] = (uint32_t*)pv; ps->pairs.vals[1] = pv ? (uint32_t*)pv + total : nullptr;
  ps->or_bits = (uint64_t*)ph;             // 8-byte aligned head of the slot
  ps->hist = (uint32_t*)((char*)ph + 8);
  ps->sums = ps->hist + ps->entries;
  HIP_TRY(ctx, hipMemsetAsync(ps->or_bits, 0, 8, s));
  return PLK_OK;
}

// `passes`: the byte passes to enqueue, from the lowest; a caller who knows that the bytes above are zero in every
// used key (so that their passes would return at once) spares their launches
int pair_sort_run(plk_ctx* ctx, hipStream_t s, const PairSort& ps, double alg_per_pass, int passes = 8) {
  // a pass whose digit is zero in every key returns at once
  for (int pass = 0; pass < passes; ++pass) {
    const std::string ptag = "/p" + std::to_string(pass);
    { ProfScope p(ctx, s, "sig_hist_kernel" + ptag, alg_per_pass / 3);
      hipLaunchKernelGGL(sig_hist_kernel, dim3(ps.waves), dim3(64), 0, s, ps.pairs, ps.total, pass, (const uint64_t*)ps.or_bits, ps.hist, ps.waves); }
    { ProfScope p(ctx, s, "sig_scan_kernel" + ptag, alg_per_pass / 3);
      hipLaunchKernelGGL(sig_scan_tiles_kernel, dim3(ps.scan_tiles), dim3(SIG_SCAN_THREADS), 0, s, ps.hist, ps.entries, ps.sums, pass,
                         (const uint64_t*)ps.or_bits);
      hipLaunchKernelGGL(sig_scan_sums_kernel, dim3(1), dim3(SIG_SCAN_THREADS), 0, s, ps.sums, ps.scan_tiles, pass, (const uint64_t*)ps.or_bits); }
    ProfScope p(ctx, s, "sig_scatter_kernel" + ptag, alg_per_pass / 3);
    hipLaunchKernelGGL(sig_scatter_kernel, dim3(ps.waves), dim3(64), 0, s, ps.pairs, ps.total, pass, (const uint64_t*)ps.or_bits,
                       (const uint32_t*)ps.hist, (const uint32_t*)ps.sums, ps.waves);
    HIP_TRY(ctx, hipGetLastError());
  }
  return PLK_OK;
}

// sigma table of n rows from node ids (device pointers), enqueued on s
int launch_sigma_table(plk_ctx* ctx, hipStream_t s, const uint64_t* const node_ids[4], size_t n, uint64_t* const sigma_out[4]) {
  if (n == 0) return PLK_OK;
  if (n > (size_t(1) << 30)) return fail(ctx, PLK_ERR_ARG, "sigma table: more than 2^30 rows");
  SigCols cols;
  for (int k = 0; k < 4; ++k) {
    if (!node_ids[k] || !sigma_out[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    cols.node[k] = node_ids[k];
    cols.sigma[k] = sigma_out[k];
  }
  const uint64_t total = 4 * (uint64_t)n;
  int rc;
  PairSort ps;
  if ((rc = pair_sort_alloc(ctx, s, "sig", total, &ps))) return rc;
  // algorithmic bytes: 4 node ids in and 4 row indices out per row, in equal shares over the 10 steps
  const double alg = (32.0 * (double)n + 32.0 * (double)n) / 10.0;
  {
    ProfScope p(ctx, s, "sig_keys_kernel", alg);
    hipLaunchKernelGGL(sig_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cols, (uint64_t)n, ps.pairs.keys[0], ps.pairs.vals[0],
                       ps.or_bits);
    HIP_TRY(ctx, hipGetLastError());
  }
  if ((rc = pair_sort_run(ctx, s, ps, alg))) return rc;
  {
    ProfScope p(ctx, s, "sig_link_kernel", alg);
    hipLaunchKernelGGL(sig_link_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ps.pairs, (const uint64_t*)ps.or_bits, total, cols);
    HIP_TRY(ctx, hipGetLastError());
  }
  return PLK_OK;
}

// ---- commitments: multi-scalar multiplication in G1 (msm.cuh)

// window width for a table of nb bases: about log2(nb) bits, so that the 2^c buckets get a few entries each even
// when only one window's worth of digits lands in them
int msm_window_bits(size_t nb) {
  int c = 4;
  while (c < MSM_MAX_WINDOW_BITS && (size_t(1) << (c + 1)) <= nb) ++c;
  return c;
}

void msm_free_table(plk_ctx* ctx) {
  if (ctx->msm_table) (void)hipFree(ctx->msm_table);
  ctx->msm_table = nullptr; ctx->msm_nbases = 0;
}

// table[w * nb + i] = 2^(offset of window w) * bases[i] for the device array `bases` (msm.cuh); replaces the context's table
int msm_build_table(plk_ctx* ctx, hipStream_t s, const uint64_t* bases, size_t nb) {
  msm_free_table(ctx);
  if (nb == 0) return PLK_OK;
  if (nb >= (size_t(1) << 26)) return fail(ctx, PLK_ERR_ARG, "msm: more than 2^26 bases");
  const MsmWindows win = msm_windows_for(msm_window_bits(nb));
  const size_t slots = nb * (size_t)win.windows;
  uint64_t *zs = nullptr, *prods = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&ctx->msm_table, slots * A28_WORDS * 4));
  hipError_t e = hipMalloc((void**)&zs, slots * 48);
  if (e == hipSuccess) e = hipMalloc((void**)&prods, slots * 48);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(msm_table_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s, bases, (uint64_t)nb, win, ctx->msm_table, zs, prods);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (zs) (void)hipFree(zs);
  if (prods) (void)hipFree(prods);
  if (e != hipSuccess) { msm_free_table(ctx); HIP_TRY(ctx, e); }
  ctx->msm_nbases = nb; ctx->msm_win = win;
  return PLK_OK;
}

// Enqueues sum_i scalars[i] * bases[i], 0 < n <= number of bases, scalars on the device (Montgomery Fr), bases = the
// context's table, on stream s; the sum arrives in *host_sum (page-locked) as a Jacobian point in the ABI form.
// `slot` names the workspace: sums in flight at the same time on different streams need different slots.
int msm_enqueue(plk_ctx* ctx, hipStream_t s, const uint64_t* scalars, size_t n, int slot, G1Jac* host_sum) {
  const std::string ws = "msm" + std::to_string(slot);
  const MsmWindows win = ctx->msm_win;
  const int c = win.c, windows = win.windows;
  const uint64_t total = (uint64_t)n * windows, nbuckets = 1ull << c;
  // lanes per bucket: about 32 entries each (measured flat between 20 and 45; their partial sums are added up by a
  // second kernel whose cost grows with the lane count)
  // (at most half of MSM_BIG_BUCKET for the average bucket: the workgroup-per-bucket kernel is for the few crowded
  // ones, and took 90 ms when half of all buckets of a 2^24-term sum sat just above its threshold)
  uint32_t lanes = (uint32_t)std::min<uint64_t>(MSM_BIG_BUCKET / 2, std::max<uint64_t>(1, (total >> c) / 32));
  // a small sum does not fill the GPU (about 2 * 10^5 lanes at three waves per SIMD): its time is the longest chain of
  // additions, so the runs are split further while lanes are free
  while (lanes * 2 <= MSM_BIG_BUCKET / 2 && nbuckets * lanes * 2 <= 196608 && (total >> c) / (lanes * 2) >= 6) lanes *= 2;
  // that was the average bucket; every bucket gets one lane per `per_lane` entries of its own run (msm_lanes_kernel)
  const uint32_t per_lane = (uint32_t)std::max<uint64_t>(1, ((total >> c) + lanes - 1) / lanes);
  const uint64_t lane_bound = total / per_lane + nbuckets;          // sum of ceil(run / per_lane) over the buckets, at most
  int rc;
  PairSort ps;
  if ((rc = pair_sort_alloc(ctx, s, ws, total, &ps, false))) return rc;     // the keys carry the table index
  void *pf, *pb, *pa;
  constexpr size_t point_bytes = J28_WORDS * 4;
  if ((rc = ws_get(ctx, ws + "_bounds", 2 * nbuckets * 4, &pf))) return rc;
  if ((rc = ws_get(ctx, ws + "_buckets", nbuckets * point_bytes, &pb))) return rc;
  void *pp_part, *po;
  if ((rc = ws_get(ctx, ws + "_partials", lane_bound * point_bytes, &pp_part))) return rc;
  if ((rc = ws_get(ctx, ws + "_lanes", (nbuckets + 1) * 4, &po))) return rc;
  const uint32_t* off = (const uint32_t*)po;
  if ((rc = ws_get(ctx, ws + "_fold", 2 * 256 * point_bytes + 2 * 144, &pa))) return rc;
  uint32_t* first = (uint32_t*)pf; uint32_t* last = first + nbuckets;
  HIP_TRY(ctx, hipMemsetAsync(pf, 0, 2 * nbuckets * 4, s));
  {
    ProfScope p(ctx, s, "msm_digits_kernel", 32.0 * (double)n);
    hipLaunchKernelGGL(msm_digits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, scalars, (uint64_t)n, (uint64_t)ctx->msm_nbases, win,
                       ps.pairs.keys[0], ps.or_bits);
    HIP_TRY(ctx, hipGetLastError());
  }
  if ((rc = pair_sort_run(ctx, s, ps, 0.0, (MSM_KEY_DIGIT_BITS + 7) / 8))) return rc;     // the OR holds digits only
  {
    ProfScope p(ctx, s, "msm_bounds_kernel", 0.0);
    hipLaunchKernelGGL(msm_bounds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ps.pairs, (const uint64_t*)ps.or_bits, total, first, last);
  }
  {
    ProfScope p(ctx, s, "msm_lanes_kernel", 8.0 * (double)nbuckets);
    hipLaunchKernelGGL(msm_lanes_kernel, dim3(1), dim3(SIG_SCAN_THREADS), 0, s, (const uint32_t*)first, (const uint32_t*)last, (uint32_t)nbuckets, per_lane,
                       (uint32_t*)po);
  }
  {
    ProfScope p(ctx, s, "msm_bucket_sum_kernel", 128.0 * (double)total);   // one affine table entry per pair
    hipLaunchKernelGGL(msm_bucket_sum_kernel, dim3((unsigned)((lane_bound + 255) / 256)), dim3(256), 0, s, ps.pairs, (const uint64_t*)ps.or_bits,
                       (const uint32_t*)first, (const uint32_t*)last, (uint32_t)nbuckets, off, (const uint32_t*)ctx->msm_table, (uint32_t*)pp_part);
  }
  {
    ProfScope p(ctx, s, "msm_bucket_combine_kernel", (double)point_bytes * (double)lane_bound);
    hipLaunchKernelGGL(msm_bucket_combine_kernel, dim3((unsigned)((2 * nbuckets + 255) / 256)), dim3(256), 0, s, (const uint32_t*)pp_part, off,
                       (uint32_t)nbuckets, (uint32_t*)pb);
    hipLaunchKernelGGL(msm_bucket_combine_big_kernel, dim3((unsigned)std::min<uint64_t>(64, (nbuckets + MSM_BIG_THREADS - 1) / MSM_BIG_THREADS)),
                       dim3(MSM_BIG_THREADS), 0, s, (const uint32_t*)pp_part, off, (uint32_t)nbuckets, (uint32_t*)pb);
  }
  // folds: 2^c points -> 2^ceil(c/2) -> ... -> 2; then sum_d d * B_d = B_1
  const uint32_t* in = (const uint32_t*)pb;
  uint32_t* pp[2] = {(uint32_t*)pa, (uint32_t*)pa + 256 * J28_WORDS};
  uint64_t* last_two = (uint64_t*)((uint32_t*)pa + 2 * 256 * J28_WORDS);      // the last fold's two points in the ABI form
  int cc = c, flip = 0;
  while (cc > 1) {
    const int next = cc - (cc >> 1);
    ProfScope p(ctx, s, "msm_fold_kernel/c" + std::to_string(cc), (double)point_bytes * (double)(1ull << cc));
    hipLaunchKernelGGL(msm_fold_kernel, dim3(1u << next), dim3(MSM_FOLD_THREADS), 0, s, in, cc, pp[flip], next == 1 ? last_two : (uint64_t*)nullptr);
    in = pp[flip]; flip ^= 1; cc = next;
  }
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(host_sum, last_two + 18, 144, hipMemcpyDeviceToHost, s));
  return PLK_OK;
}

// Jacobian -> affine on the host with 64-bit limbs (the 32-bit code of g1.cuh, written for the GPU, takes ~0.15 ms
// for the inversion on a host core; this takes ~0.02 ms).  Same Montgomery form: pairs of 32-bit limbs.
struct Fq64 { uint64_t l[6]; };
inline Fq64 fq64_from(const Fq& a) { Fq64 r; for (int i = 0; i < 6; ++i) r.l[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32); return r; }
inline Fq fq64_to(const Fq64& a) { Fq r; for (int i = 0; i < 6; ++i) { r.l[2 * i] = (uint32_t)a.l[i]; r.l[2 * i + 1] = (uint32_t)(a.l[i] >> 32); } return r; }
inline Fq64 fq64_mul(const Fq64& a, const Fq64& b) {
  static const Fq64 q = [] { Fq t; for (int i = 0; i < 12; ++i) t.l[i] = FqP::P[i]; return fq64_from(t); }();
  static const uint64_t inv = ((uint64_t)0x89f3fffcu << 32) | FqP::INV32;      // -1 / q mod 2^64 (checked below)
  typedef unsigned __int128 u128;
  uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 6; ++i) {
    u128 c = 0;
    for (int j = 0; j < 6; ++j) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[6] = (uint64_t)c; t[7] = (uint64_t)(c >> 64);
    const uint64_t m = t[0] * inv;
    c = ((u128)m * q.l[0] + t[0]) >> 64;
    for (int j = 1; j < 6; ++j) { c += (u128)m * q.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[6]; t[5] = (uint64_t)c; t[6] = t[7] + (uint64_t)(c >> 64);
  }
  // t < 2q: one conditional subtraction
  uint64_t d[6]; u128 bo = 0;
  for (int j = 0; j < 6; ++j) { const u128 x = (u128)t[j] - q.l[j] - (uint64_t)bo; d[j] = (uint64_t)x; bo = (x >> 64) & 1; }
  Fq64 r;
  for (int j = 0; j < 6; ++j) r.l[j] = bo ? t[j] : d[j];
  return r;
}
G1Affine g1_to_affine_host(const G1Jac& p) {
  G1Affine r;
  if (g1_is_inf(p)) { r.x = fq_zero(); r.y = fq_zero(); return r; }
  const Fq64 z = fq64_from(p.z);
  Fq64 acc = fq64_from(fq_one());
  uint32_t e[12];
  for (int i = 0; i < 12; ++i) e[i] = FqP::P[i];
  e[0] -= 2;                                     // z^(q - 2) = 1 / z
  for (int bit = 380; bit >= 0; --bit) {
    acc = fq64_mul(acc, acc);
    if ((e[bit >> 5] >> (bit & 31)) & 1) acc = fq64_mul(acc, z);
  }
  const Fq64 zi2 = fq64_mul(acc, acc);
  r.x = fq64_to(fq64_mul(fq64_from(p.x), zi2));
  r.y = fq64_to(fq64_mul(fq64_from(p.y), fq64_mul(zi2, acc)));
  return r;
}

// `count` sums over the same bases, scalars[k] on the device with n elements each; the affine results (x, y; (0, 0) =
// infinity) go to out_xy[12 k ...] in host memory.  Waits for the stream.  With more than one sum they alternate
// between s and the context's side stream, so that one sum's latency-bound bucket reduction runs beside the next
// one's additions.
int launch_msm(plk_ctx* ctx, hipStream_t s, const uint64_t* const* scalars, size_t count, size_t n, uint64_t* out_xy) {
  if (!scalars || !out_xy) return fail(ctx, PLK_ERR_ARG, "null pointer");
  for (size_t k = 0; k < count; ++k) if (!scalars[k] && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (n > ctx->msm_nbases) return fail(ctx, PLK_ERR_LENGTH, "more scalars than bases set with plk_msm_set_bases");
  for (size_t i = 0; i < 12 * count; ++i) out_xy[i] = 0;
  if (n == 0 || count == 0) return PLK_OK;
  if (ctx->msm_sums_cap < count) {
    if (ctx->msm_sums) HIP_TRY(ctx, hipHostFree(ctx->msm_sums));
    ctx->msm_sums = nullptr; ctx->msm_sums_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->msm_sums, std::max<size_t>(count, 8) * sizeof(G1Jac), hipHostMallocDefault));
    ctx->msm_sums_cap = std::max<size_t>(count, 8);
  }
  int rc = PLK_OK;
  const bool two = count > 1 && s != ctx->aux;
  if (two) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, s));              // the scalars are ready where s stands now
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ev_fork, 0));
  }
  for (size_t k = 0; k < count && rc == PLK_OK; ++k) {
    const int slot = two ? (int)(k & 1) : 0;
    rc = msm_enqueue(ctx, slot ? ctx->aux : s, scalars[k], n, slot, ctx->msm_sums + k);
  }
  if (two) {                                                    // also on an error: s must not run ahead of the side stream
    (void)hipEventRecord(ctx->ev_join, ctx->aux);
    (void)hipStreamWaitEvent(s, ctx->ev_join, 0);
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(s));
  for (size_t k = 0; k < count; ++k) {
    const G1Affine r = g1_to_affine_host(ctx->msm_sums[k]);      // one base-field inversion each, on the host
    for (int i = 0; i < 6; ++i) {
      out_xy[12 * k + i] = (uint64_t)r.x.l[2 * i] | ((uint64_t)r.x.l[2 * i + 1] << 32);
      out_xy[12 * k + 6 + i] = (uint64_t)r.y.l[2 * i] | ((uint64_t)r.y.l[2 * i + 1] << 32);
    }
  }
  return PLK_OK;
}

int fetch_status(plk_ctx* ctx, hipStream_t s) {
  HIP_TRY(ctx, hipMemcpyAsync(ctx->status_host, ctx->status_dev, sizeof(GpStatus), hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  const GpStatus st = *ctx->status_host;
  int rc = PLK_OK;
  // the reference's loop reaches the slice index of row j before that row's inversion
  if (st.bad_sigma_row != ~0ull && st.bad_sigma_row <= st.zero_den_row) {
    ctx->last_row = (long long)st.bad_sigma_row;
    rc = fail(ctx, PLK_ERR_SIGMA_RANGE, "sigma index out of range at row " + std::to_string(st.bad_sigma_row));
  } else if (st.zero_den_row != ~0ull) {
    ctx->last_row = (long long)st.zero_den_row;
    rc = fail(ctx, PLK_ERR_ZERO_DENOMINATOR, "zero denominator at row " + std::to_string(st.zero_den_row));
  }
  if (rc != PLK_OK) { int r2 = status_reset(ctx, s); if (r2) return r2; }
  return rc;
}

// ---- host staging helpers

// Device copies of a host entry point's arguments and results.  The buffers are slots of the
// context's grow-only workspace ("stage0", "stage1", ... in order of use), so a repeated call of the
// same shape allocates nothing; they are valid until the next host entry point on this context.
bool host_is_pageable(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return true; }   // unknown to the runtime
  return a.type == hipMemoryTypeUnregistered;
}

int copy_lanes_init(plk_ctx* ctx) {
  if (ctx->copy_ready) return PLK_OK;
  for (auto& lane : ctx->copy_lanes) {
    for (int s = 0; s < 2; ++s) {
      if (!lane.slot[s]) HIP_TRY(ctx, hipHostMalloc(&lane.slot[s], COPY_CHUNK, hipHostMallocDefault));
      if (!lane.ev[s]) HIP_TRY(ctx, hipEventCreateWithFlags(&lane.ev[s], hipEventDisableTiming));
    }
    if (!lane.st) HIP_TRY(ctx, hipStreamCreateWithFlags(&lane.st, hipStreamNonBlocking));
  }
  ctx->copy_pool.start();
  ctx->copy_ready = true;
  return PLK_OK;
}

// dst <- src, one side pageable host memory, the other device memory; returns when the bytes have arrived
int threaded_copy(plk_ctx* ctx, void* dst, const void* src, size_t bytes, bool to_device) {
  int rc = copy_lanes_init(ctx);
  if (rc) return rc;
  const size_t chunks = (bytes + COPY_CHUNK - 1) / COPY_CHUNK;
  const size_t stride = (size_t)ctx->copy_pool.workers;
  std::atomic<int> failed{0};
  auto span = [&](size_t c, size_t* off) { *off = c * COPY_CHUNK; return std::min(COPY_CHUNK, bytes - *off); };
  auto body = [&](int w) {
    CopyLane& lane = ctx->copy_lanes[w];
    bool ok = hipSetDevice(ctx->device) == hipSuccess;
    int k = 0;
    if (to_device) {
      for (size_t c = (size_t)w; ok && c < chunks; c += stride, ++k) {
        const int s = k & 1;
        size_t off; const size_t len = span(c, &off);
        if (k >= 2) ok = hipEventSynchronize(lane.ev[s]) == hipSuccess;            // the slot's previous DMA has read it
        if (!ok) break;
        memcpy(lane.slot[s], (const char*)src + off, len);
        ok = hipMemcpyAsync((char*)dst + off, lane.slot[s], len, hipMemcpyHostToDevice, lane.st) == hipSuccess &&
             hipEventRecord(lane.ev[s], lane.st) == hipSuccess;
      }
    } else {
      auto fetch = [&](size_t c, int s) {
        size_t off; const size_t len = span(c, &off);
        return hipMemcpyAsync(lane.slot[s], (const char*)src + off, len, hipMemcpyDeviceToHost, lane.st) == hipSuccess &&
               hipEventRecord(lane.ev[s], lane.st) == hipSuccess;
      };
      if ((size_t)w < chunks) ok = ok && fetch((size_t)w, 0);
      for (size_t c = (size_t)w; ok && c < chunks; c += stride, ++k) {        // DMA of the next chunk runs beside this memcpy
        const int s = k & 1;
        if (c + stride < chunks) ok = fetch(c + stride, s ^ 1);
        ok = ok && hipEventSynchronize(lane.ev[s]) == hipSuccess;
        if (!ok) break;
        size_t off; const size_t len = span(c, &off);
        memcpy((char*)dst + off, lane.slot[s], len);
      }
    }
    if (hipStreamSynchronize(lane.st) != hipSuccess) ok = false;
    if (!ok) failed = 1;
  };
  ctx->copy_pool.run(body);
  if (failed) { (void)hipGetLastError(); return fail(ctx, PLK_ERR_HIP, "copy between host and device memory failed"); }
  return PLK_OK;
}

// host -> device on the context's stream (the staging slot is fresh, nothing enqueued touches it)
int upload_bytes(plk_ctx* ctx, void* dst, const void* host, size_t bytes) {
  if (bytes >= COPY_MIN && host_is_pageable(host)) return threaded_copy(ctx, dst, host, bytes, true);
  HIP_TRY(ctx, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  return PLK_OK;
}

// device -> host behind everything enqueued on the context's stream; the caller synchronises the stream afterwards
int download(plk_ctx* ctx, void* host, const void* src, size_t bytes) {
  if (bytes >= COPY_MIN && host_is_pageable(host)) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return threaded_copy(ctx, host, src, bytes, false);
  }
  HIP_TRY(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return PLK_OK;
}

// Staging buffers of a host entry point.  Going out of scope waits for the context's stream: copies from and to the
// CALLER's arrays are asynchronous when those are page-locked, and an entry point that returns early with an error
// must not leave one in flight behind it (after a successful call the stream is idle already and the wait is free).
struct Staged {
  plk_ctx* ctx;
  int used = 0;
  explicit Staged(plk_ctx* c) : ctx(c) {}
  Staged(const Staged&) = delete;
  Staged& operator=(const Staged&) = delete;
  ~Staged() { (void)hipStreamSynchronize(ctx->stream); }
  int alloc(size_t bytes, void** out) {
    return ws_get(ctx, "stage" + std::to_string(used++), std::max<size_t>(bytes, 32), out);
  }
  int upload(const void* host, size_t bytes, void** out) {
    int rc = alloc(bytes, out);
    if (rc) return rc;
    return bytes ? upload_bytes(ctx, *out, host, bytes) : PLK_OK;
  }
};

int use_device(plk_ctx* ctx) {
  if (!ctx) return PLK_ERR_ARG;
  ctx->last_error.clear();
  ctx->last_row = -1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return PLK_OK;
}

// plk_round23 on the 4n-point domain with uploads, kernels and downloads overlapped.  The columns go up in the
// order wires, selectors (three at a time), sigma; every group's transforms start when the group has arrived
// (page-locked arrays: DMA on a copy stream, the kernels wait for its event; pageable arrays: the threaded copy
// returns when the group is resident).  The gate constraints need wires and selectors only, so they run - and
// their vector goes down - while sigma is still on its way up and being transformed; z goes down as soon as the
// grand product has finished.  What remains exposed is the upload of the 13 + 4 columns and, at the end, the
// permutation kernel and the download of its vector.
int round23_host_pipelined(plk_ctx* ctx, const uint64_t* const wires[4], const uint64_t* const sel[9], const uint64_t* const sigma[4],
                           size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* z_out,
                           uint64_t* gate_out, uint64_t* perm_out) {
  int rc;
  hipStream_t s = ctx->stream, sa = ctx->aux, up = ctx->copy_up, down = ctx->copy_down;
  enum { EV_W = 0, EV_S, EV_Z, EV_GATE, EV_PERM, EV_SEL0 };   // EV_SEL0 .. EV_SEL0 + 2
  // every way out of this function - errors included - leaves no copy from or to the caller's arrays in flight
  // (declared before `st`: destroyed after it, so the context's own stream has been waited for by then)
  struct Drain {
    plk_ctx* c;
    ~Drain() { (void)hipStreamSynchronize(c->aux); (void)hipStreamSynchronize(c->copy_up); (void)hipStreamSynchronize(c->copy_down); }
  } drain{ctx};
  Staged st(ctx);
  void *dw[4], *dsg[4], *dsel[9] = {nullptr}, *dz, *dg = nullptr, *dp = nullptr;
  for (int k = 0; k < 4; ++k) { if ((rc = st.alloc(n * ELEM, &dw[k]))) return rc; if ((rc = st.alloc(n * 8, &dsg[k]))) return rc; }
  for (int k = 0; k < 9; ++k) if (sel && sel[k]) { if ((rc = st.alloc(n * ELEM, &dsel[k]))) return rc; }
  if ((rc = st.alloc((n + 1) * ELEM, &dz))) return rc;
  if (gate_out && (rc = st.alloc(N * ELEM, &dg))) return rc;
  if (perm_out && (rc = st.alloc(N * ELEM, &dp))) return rc;
  R23Ws ws;
  if ((rc = r23_workspace(ctx, n, N, (uint64_t*)dz, &ws))) return rc;
  // the copy streams start behind whatever the context's stream has been given so far (the call guard has ordered
  // that behind the context's earlier calls): the staging buffers may still be read by them
  HIP_TRY(ctx, hipEventRecord(ctx->ev_pipe[EV_PERM], s));
  HIP_TRY(ctx, hipStreamWaitEvent(up, ctx->ev_pipe[EV_PERM], 0));
  HIP_TRY(ctx, hipStreamWaitEvent(down, ctx->ev_pipe[EV_PERM], 0));

  auto put = [&](void* dst, const void* host, size_t bytes) -> int {
    if (bytes >= COPY_MIN && host_is_pageable(host)) {
      HIP_TRY(ctx, hipStreamSynchronize(up));                 // the threaded copy uses streams of its own
      return threaded_copy(ctx, dst, host, bytes, true);
    }
    HIP_TRY(ctx, hipMemcpyAsync(dst, host, bytes, hipMemcpyHostToDevice, up));
    return PLK_OK;
  };
  auto arrived = [&](int ev, hipStream_t waiter, hipStream_t waiter2 = nullptr) -> int {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_pipe[ev], up));
    HIP_TRY(ctx, hipStreamWaitEvent(waiter, ctx->ev_pipe[ev], 0));
    if (waiter2) HIP_TRY(ctx, hipStreamWaitEvent(waiter2, ctx->ev_pipe[ev], 0));
    return PLK_OK;
  };
  // page-locked destination: DMA on the download stream behind `ev`; pageable: wait for `ev`, then the threaded copy
  auto get = [&](void* host, const void* src, size_t bytes, int ev, bool deferred) -> int {
    const bool threaded = bytes >= COPY_MIN && host_is_pageable(host);
    if (threaded != deferred) return PLK_OK;                  // each kind of destination is served in its own phase
    if (threaded) {
      HIP_TRY(ctx, hipEventSynchronize(ctx->ev_pipe[ev]));
      return threaded_copy(ctx, host, src, bytes, false);
    }
    HIP_TRY(ctx, hipStreamWaitEvent(down, ctx->ev_pipe[ev], 0));
    HIP_TRY(ctx, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, down));
    return PLK_OK;
  };

  const uint64_t* w[4]; const uint64_t* sg[4]; const uint64_t* sl[9] = {nullptr};
  for (int k = 0; k < 4; ++k) { w[k] = (const uint64_t*)dw[k]; sg[k] = (const uint64_t*)dsg[k]; }
  for (int k = 0; k < 9; ++k) sl[k] = (const uint64_t*)dsel[k];

  // 1. wires
  for (int k = 0; k < 4; ++k) if ((rc = put(dw[k], wires[k], n * ELEM))) return rc;
  if ((rc = arrived(EV_W, s, sa))) return rc;
  {
    const uint32_t kinds[4] = {NTT_IN_FIELD, NTT_IN_FIELD, NTT_IN_FIELD, NTT_IN_FIELD};
    const int col_of[4] = {0, 1, 2, 3};
    if ((rc = r23_extend_group(ctx, s, ws, w, kinds, col_of, 4))) return rc;
  }
  // 2. selectors, three at a time
  {
    const uint64_t* in[3]; uint32_t kinds[3]; int col_of[3]; size_t m = 0; int group = 0;
    auto flush = [&]() -> int {
      if (m == 0) return PLK_OK;
      int r = arrived(EV_SEL0 + (group % 3), s);
      if (r) return r;
      ++group;
      r = r23_extend_group(ctx, s, ws, in, kinds, col_of, m);
      m = 0;
      return r;
    };
    for (int k = 0; k < 9; ++k) {
      if (!dsel[k]) continue;
      if ((rc = put(dsel[k], sel[k], n * ELEM))) return rc;
      in[m] = sl[k]; kinds[m] = NTT_IN_FIELD; col_of[m] = 4 + k; ++m;
      if (m == 3 && (rc = flush())) return rc;
    }
    if ((rc = flush())) return rc;
  }
  // 3. gate constraints, and their vector on its way down at once
  if (gate_out) {
    if ((rc = r23_gate(ctx, s, ws, w, sl, (uint64_t*)dg))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_pipe[EV_GATE], s));
    if ((rc = get(gate_out, dg, N * ELEM, EV_GATE, false))) return rc;
  }
  // 4. sigma; the grand-product chain on the side stream, the sigma transforms on the main one
  for (int k = 0; k < 4; ++k) if ((rc = put(dsg[k], sigma[k], n * 8))) return rc;
  if ((rc = arrived(EV_S, s, sa))) return rc;
  {
    SideJoin side{ctx, s, sa};
    if ((rc = r23_z_chain(ctx, sa, ws, w, sg, beta, gamma, ctx->ev_pipe[EV_Z]))) return rc;
    side.record();
    if (z_out && (rc = get(z_out, dz, (n + 1) * ELEM, EV_Z, false))) return rc;
    const uint32_t kinds[4] = {NTT_IN_U64, NTT_IN_U64, NTT_IN_U64, NTT_IN_U64};
    const int col_of[4] = {13, 14, 15, 16};
    const SigmaScale sg_scale(ws, beta, 0);
    if ((rc = r23_extend_group(ctx, s, ws, sg, kinds, col_of, 4, "ntt_tmp", &sg_scale.opt))) return rc;
    side.join();
  }
  if (perm_out) {
    if ((rc = r23_perm(ctx, s, ws, w, sg, beta, gamma, z_shift, (uint64_t*)dp))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_pipe[EV_PERM], s));
    if ((rc = get(perm_out, dp, N * ELEM, EV_PERM, false))) return rc;
  }
  // 5. pageable destinations: the calling thread copies them as their kernels finish
  if (gate_out && (rc = get(gate_out, dg, N * ELEM, EV_GATE, true))) return rc;
  if (z_out && (rc = get(z_out, dz, (n + 1) * ELEM, EV_Z, true))) return rc;
  if (perm_out && (rc = get(perm_out, dp, N * ELEM, EV_PERM, true))) return rc;
  // 6. deferred status (waits for the main stream), then the downloads still in flight
  rc = fetch_status(ctx, s);
  HIP_TRY(ctx, hipStreamSynchronize(down));
  HIP_TRY(ctx, hipStreamSynchronize(up));
  return rc;
}

// Every entry point that enqueues on a stream holds one of these for the length of the call.  The workspace, the
// status word and the side stream are shared by every call on the context, so calls are ordered one after another
// whatever stream they are given: the call's last act (destructor) records ev_order on ITS OWN stream - a handle
// that is certainly alive, the caller being inside the call - and the next call, if it comes on another stream,
// makes that stream wait for the event.  The library never touches a caller's stream handle after the call that
// brought it has returned, so a caller may destroy a stream once the work it gave it has finished.
// A stream under HIP graph capture is left out of this protocol (an event recorded into a capture cannot be waited
// for from outside it, and the wait would pull the other stream into the capture): the captured launches are
// ordered by the graph, and the caller orders the graph's replays against other calls on the context.
struct StreamCall {
  plk_ctx* ctx;
  hipStream_t s;
  bool capturing = false;
  int rc;
  StreamCall(plk_ctx* c, hipStream_t st) : ctx(c), s(st) { rc = begin(); }
  StreamCall(const StreamCall&) = delete;
  StreamCall& operator=(const StreamCall&) = delete;
  int begin() {
    int r = use_device(ctx);
    if (r) return r;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    if (cs == hipStreamCaptureStatusActive) { capturing = true; return PLK_OK; }
    if (ctx->have_order && ctx->order_stream != s) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_order, 0));
    return PLK_OK;
  }
  ~StreamCall() {
    if (rc != PLK_OK || capturing) return;
    if (hipEventRecord(ctx->ev_order, s) == hipSuccess) {
      ctx->have_order = true;
      ctx->order_stream = s;           // compared by value only, never used as a stream again
    } else {                           // cannot leave a marker: wait here, so that the next call has nothing to wait for
      (void)hipGetLastError();
      (void)hipDeviceSynchronize();
      ctx->have_order = false;
    }
  }
};

// `STREAM_CALL(ctx, stream)` opens an entry point that enqueues on the caller's stream, `HOST_CALL(ctx)` one that
// enqueues on the context's own.
#define STREAM_CALL(ctx, stream)                     \
  StreamCall call_(ctx, (hipStream_t)(stream));      \
  if (call_.rc) return call_.rc
#define HOST_CALL(ctx)                               \
  if (!(ctx)) return PLK_ERR_ARG;                    \
  StreamCall call_(ctx, (ctx)->stream);              \
  if (call_.rc) return call_.rc

}  // namespace

// ============================================================================ extern "C"

extern "C" {

int plk_ctx_create(int device, plk_ctx** out) {
  if (!out) return PLK_ERR_ARG;
  *out = nullptr;
  int count = 0;
  hipError_t ec = hipGetDeviceCount(&count);
  if (ec != hipSuccess || device < 0 || device >= count) {
    fprintf(stderr, "plk_ctx_create: device %d not available (hipGetDeviceCount: %s, count %d)\n", device,
            hipGetErrorString(ec), count);
    return PLK_ERR_HIP;
  }
  plk_ctx* ctx = new (std::nothrow) plk_ctx();
  if (ctx) ctx->field_id = PLK_FIELD_ID;
  if (!ctx) return PLK_ERR_NOMEM;
  if (const char* e = getenv("PLONK_NTT_FIRST_RADIX_LOG")) ctx->first_radix_log = atoi(e);
  if (const char* e = getenv("PLONK_HIP_FUSED_MIN_LOG")) ctx->fused_constraints_min_log = atoi(e);
  if (const char* e = getenv("PLONK_NTT_TWO_PASS_MAX_LOG")) ctx->two_pass_max_log = std::max(12, std::min(atoi(e), 2 * NTT_TILE_LOG));   // measurement knob
  ctx->device = device;
#define CTX_TRY(expr)                                                                           \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "plk_ctx_create: %s failed: %s\n", #expr, hipGetErrorString(e_));        \
      plk_ctx_destroy(ctx);                                                                     \
      return PLK_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)
  CTX_TRY(hipSetDevice(device));
  CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  CTX_TRY(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
  CTX_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  CTX_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  CTX_TRY(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
  CTX_TRY(hipStreamCreateWithFlags(&ctx->copy_up, hipStreamNonBlocking));
  CTX_TRY(hipStreamCreateWithFlags(&ctx->copy_down, hipStreamNonBlocking));
  for (auto& e : ctx->ev_pipe) CTX_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CTX_TRY(hipMalloc((void**)&ctx->status_dev, sizeof(GpStatus)));
  CTX_TRY(hipHostMalloc((void**)&ctx->status_host, sizeof(GpStatus)));
  CTX_TRY(hipMemset(ctx->status_dev, 0xff, sizeof(GpStatus)));
  // this device's copy of the fixed scratch array (the device is current: hipSetDevice above); loading the kernels of
  // the entry point that uses it now keeps its first call free of the runtime's own lazy set-up
  CTX_TRY(hipGetSymbolAddress((void**)&ctx->noalloc_ptr, HIP_SYMBOL(scratch)));
  {
    hipFuncAttributes fa;
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_local_z_kernel));
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_ratio_scan_kernel));
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_apply_kernel<GP_ITEMS>));
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_apply_small_kernel));
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_apply_fused_kernel<GP_ITEMS>));
    CTX_TRY(hipFuncGetAttributes(&fa, (const void*)gp_local_z_small_kernel));
  }
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<true, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<true, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_final_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_final_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_fs_kernel<FS_A_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_fs_kernel<FS_A_LAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_fs_kernel<FS_B_FIRST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_fs_kernel<FS_B_LAST_NATURAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
  CTX_TRY(hipFuncSetAttribute((const void*)ntt_fs_kernel<FS_B_LAST_TRANSPOSED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)NTT_LDS_BYTES));
#undef CTX_TRY
  *out = ctx;
  return PLK_OK;
}

void plk_ctx_destroy(plk_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  (void)hipDeviceSynchronize();
  for (auto& kv : ctx->plans) free_plan_tables(kv.second);
  for (auto& b : ctx->master_sh) if (b.p) (void)hipFree(b.p);
  for (auto& kv : ctx->fs_plans) for (DevBuf* b : {&kv.second.sub_a, &kv.second.sub_b, &kv.second.block, &kv.second.hi, &kv.second.lo}) if (b->p) (void)hipFree(b->p);
  for (auto& kv : ctx->ws) if (kv.second.p) (void)hipFree(kv.second.p);
  msm_free_table(ctx);
  if (ctx->msm_sums) (void)hipHostFree(ctx->msm_sums);
  ctx->copy_pool.shutdown();
  for (auto& lane : ctx->copy_lanes) {
    for (int s = 0; s < 2; ++s) { if (lane.slot[s]) (void)hipHostFree(lane.slot[s]); if (lane.ev[s]) (void)hipEventDestroy(lane.ev[s]); }
    if (lane.st) (void)hipStreamDestroy(lane.st);
  }
  for (auto& kv : ctx->powers) { if (kv.second.lo.p) (void)hipFree(kv.second.lo.p); if (kv.second.hi.p) (void)hipFree(kv.second.hi.p); }
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto& e : ctx->event_pool) (void)hipEventDestroy(e);
  if (ctx->status_dev) (void)hipFree(ctx->status_dev);
  if (ctx->status_host) (void)hipHostFree(ctx->status_host);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_order) (void)hipEventDestroy(ctx->ev_order);
  for (auto& e : ctx->ev_pipe) if (e) (void)hipEventDestroy(e);
  if (ctx->copy_up) (void)hipStreamDestroy(ctx->copy_up);
  if (ctx->copy_down) (void)hipStreamDestroy(ctx->copy_down);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int plk_host_alloc(plk_ctx* ctx, size_t bytes, void** out) {
  int rc = use_device(ctx);
  if (rc) return rc;
  if (!out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  *out = nullptr;
  HIP_TRY(ctx, hipHostMalloc(out, std::max<size_t>(bytes, 64), hipHostMallocDefault));
  return PLK_OK;
}

void plk_host_free(plk_ctx* ctx, void* p) {
  // the context is not touched: page-locked memory is freed without a current device, and a buffer may outlive
  // the context it was allocated through (a finalizer that runs after plk_ctx_destroy)
  (void)ctx;
  if (p) (void)hipHostFree(p);
}

const char* plk_strerror(int code) {
  switch (code) {
    case PLK_OK: return "ok";
    case PLK_ERR_LENGTH: return "Evaluation length must match the domain size";
    case PLK_ERR_NOT_POW2: return "domain size is not a power of two";
    case PLK_ERR_TWO_ADICITY: return "domain size exceeds the field's two-adicity";
    case PLK_ERR_ZERO_DENOMINATOR: return "zero denominator in the grand product (inverse of zero)";
    case PLK_ERR_SIGMA_RANGE: return "sigma index out of range";
    case PLK_ERR_HIP: return "HIP runtime error";
    case PLK_ERR_ARG: return "invalid argument";
    case PLK_ERR_NOMEM: return "out of memory";
    default: return "unknown error";
  }
}

const char* plk_last_error(const plk_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }
long long plk_last_error_row(const plk_ctx* ctx) { return ctx ? ctx->last_row : -1; }

void plk_field_constants(uint64_t modulus[4], uint64_t one[4], uint64_t r2[4]) {
  for (int i = 0; i < 4; ++i) {
    if (modulus) modulus[i] = (uint64_t)Field::P[2 * i] | ((uint64_t)Field::P[2 * i + 1] << 32);
    if (one) one[i] = (uint64_t)Field::ONE[2 * i] | ((uint64_t)Field::ONE[2 * i + 1] << 32);
    if (r2) r2[i] = (uint64_t)Field::R2[2 * i] | ((uint64_t)Field::R2[2 * i + 1] << 32);
  }
}


// ---------------------------------------------------------------------------- device memory for callers without HIP

int plk_dev_alloc(plk_ctx* ctx, size_t bytes, void** out) {
  int rc = use_device(ctx);
  if (rc) return rc;
  if (!out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  *out = nullptr;
  HIP_TRY(ctx, hipMalloc(out, std::max<size_t>(bytes, 256)));
  return PLK_OK;
}

void plk_dev_free(plk_ctx* ctx, void* p) {
  if (!ctx || !p) return;
  (void)hipSetDevice(ctx->device);
  (void)hipFree(p);          // (waits for the device: nothing enqueued can still use the block)
}

void* plk_ctx_stream(plk_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int plk_dev_upload(plk_ctx* ctx, void* stream, void* dst_dev, const void* src_host, size_t bytes) {
  STREAM_CALL(ctx, stream);
  if (bytes == 0) return PLK_OK;
  if (!dst_dev || !src_host) return fail(ctx, PLK_ERR_ARG, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (bytes >= COPY_MIN && host_is_pageable(src_host)) {
    HIP_TRY(ctx, hipStreamSynchronize(s));            // the threaded copy runs on streams of its own
    return threaded_copy(ctx, dst_dev, src_host, bytes, true);
  }
  HIP_TRY(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return PLK_OK;
}

int plk_dev_download(plk_ctx* ctx, void* stream, void* dst_host, const void* src_dev, size_t bytes) {
  STREAM_CALL(ctx, stream);
  if (bytes == 0) return PLK_OK;
  if (!dst_host || !src_dev) return fail(ctx, PLK_ERR_ARG, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (bytes >= COPY_MIN && host_is_pageable(dst_host)) {
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return threaded_copy(ctx, dst_host, src_dev, bytes, false);
  }
  HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(ctx, hipStreamSynchronize(s));
  return PLK_OK;
}

// ---------------------------------------------------------------------------- profiling

void plk_profile_begin(plk_ctx* ctx) {
  if (!ctx) return;
  for (auto& r : ctx->prof) { ctx->event_pool.push_back(r.e0); ctx->event_pool.push_back(r.e1); }
  ctx->prof.clear();
  ctx->profiling = true;
}

const char* plk_profile_end(plk_ctx* ctx) {
  if (!ctx) return "[]";
  ctx->profiling = false;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  struct Agg { size_t launches = 0; double ms = 0, bytes = 0, products = 0; };
  std::vector<std::string> order;
  std::map<std::string, Agg> agg;
  for (auto& r : ctx->prof) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = 0;
    if (!agg.count(r.label)) order.push_back(r.label);
    Agg& a = agg[r.label];
    a.launches += 1; a.ms += ms; a.bytes += r.alg_bytes; a.products += r.products;
    ctx->event_pool.push_back(r.e0); ctx->event_pool.push_back(r.e1);
  }
  ctx->prof.clear();
  std::string out = "[";
  char buf[512];
  for (size_t i = 0; i < order.size(); ++i) {
    const Agg& a = agg[order[i]];
    snprintf(buf, sizeof buf, "%s{\"label\": \"%s\", \"launches\": %zu, \"total_ms\": %.6f, \"alg_bytes\": %.1f, \"products\": %.1f}",
             i ? ", " : "", order[i].c_str(), a.launches, a.ms, a.bytes, a.products);
    out += buf;
  }
  out += "]";
  ctx->prof_report = out;
  return ctx->prof_report.c_str();
}

// ---------------------------------------------------------------------------- device API

int plk_dev_ntt_batch(plk_ctx* ctx, void* stream, const uint64_t* const* in, const uint32_t* kinds, size_t batch,
                      size_t n_in, size_t N, int inverse, uint64_t* const* out) {
  STREAM_CALL(ctx, stream);
  int rc;
  if (!in || !out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  for (size_t done = 0; done < batch; done += NTT_MAX_BATCH) {
    const size_t m = std::min<size_t>(NTT_MAX_BATCH, batch - done);
    if ((rc = launch_ntt(ctx, (hipStream_t)stream, in + done, kinds ? kinds + done : nullptr, m, n_in, N, inverse != 0, out + done)))
      return rc;
  }
  return PLK_OK;
}

int plk_dev_grand_product(plk_ctx* ctx, void* stream, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n,
                          const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out) {
  STREAM_CALL(ctx, stream);
  if (!wires || !sigma) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_grand_product(ctx, (hipStream_t)stream, wires, sigma, n, beta, gamma, z_out);
}

int plk_dev_grand_product_noalloc(plk_ctx* ctx, void* stream, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n,
                                  const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out) {
  STREAM_CALL(ctx, stream);
  if (!wires || !sigma) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  if (n > NOALLOC_MAX_ROWS)
    return fail(ctx, PLK_ERR_ARG, "plk_dev_grand_product_noalloc takes at most " + std::to_string(NOALLOC_MAX_ROWS) + " rows, got " +
                                      std::to_string(n) + ": use plk_dev_grand_product");
  if (ctx->profiling) return fail(ctx, PLK_ERR_ARG, "plk_dev_grand_product_noalloc is not available between plk_profile_begin/end (events are allocated there)");
  // the fixed array holds the block ratios, one element per block of GP_BLOCK_ROWS rows; every word a kernel reads
  // there was written by an earlier kernel of the same call, so nothing has to be cleared between calls
  return launch_grand_product(ctx, (hipStream_t)stream, wires, sigma, n, beta, gamma, z_out, (uint64_t*)ctx->noalloc_ptr);
}

int plk_dev_grand_product_local(plk_ctx* ctx, void* stream, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n,
                                size_t first, size_t rows, const uint64_t beta[4], const uint64_t gamma[4], uint64_t* totals_out) {
  STREAM_CALL(ctx, stream);
  if (!wires || !sigma || !totals_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  return gp_local(ctx, (hipStream_t)stream, wires, sigma, n, first, rows, beta, gamma, totals_out);
}

int plk_dev_grand_product_finish(plk_ctx* ctx, void* stream, size_t rows, const uint64_t* totals_all, size_t ranges, size_t mine,
                                 uint64_t* z_after_out) {
  STREAM_CALL(ctx, stream);
  int rc;
  if (rows == 0 || rows != ctx->gp_rows) return fail(ctx, PLK_ERR_ARG, "rows differ from the preceding plk_dev_grand_product_local");
  if ((rc = gp_apply_carries(ctx, (hipStream_t)stream, totals_all, ranges, mine))) return rc;
  return gp_finish(ctx, (hipStream_t)stream, z_after_out, nullptr);
}

int plk_dev_msm_set_bases(plk_ctx* ctx, const uint64_t* bases_dev, size_t n) {
  HOST_CALL(ctx);
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!bases_dev && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
  HIP_TRY(ctx, hipDeviceSynchronize());          // the caller's table may have been written on any stream
  return msm_build_table(ctx, ctx->stream, bases_dev, n);
}

int plk_dev_msm_g1(plk_ctx* ctx, void* stream, const uint64_t* scalars_dev, size_t n, uint64_t out_xy[12]) {
  STREAM_CALL(ctx, stream);
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!scalars_dev && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
  const uint64_t* one[1] = {scalars_dev};
  return launch_msm(ctx, (hipStream_t)stream, one, 1, n, out_xy);
}

int plk_dev_msm_g1_batch(plk_ctx* ctx, void* stream, const uint64_t* const scalars_dev[], size_t count, size_t n, uint64_t* out_xy) {
  STREAM_CALL(ctx, stream);
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  return launch_msm(ctx, (hipStream_t)stream, scalars_dev, count, n, out_xy);
}

int plk_dev_g1_multiples(plk_ctx* ctx, void* stream, const uint64_t base_xy[12], uint64_t first, size_t count, uint64_t* out_dev) {
  STREAM_CALL(ctx, stream);
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!base_xy || (!out_dev && count)) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (count == 0) return PLK_OK;
  G1Affine b;
  for (int i = 0; i < 6; ++i) {
    b.x.l[2 * i] = (uint32_t)base_xy[i]; b.x.l[2 * i + 1] = (uint32_t)(base_xy[i] >> 32);
    b.y.l[2 * i] = (uint32_t)base_xy[6 + i]; b.y.l[2 * i + 1] = (uint32_t)(base_xy[6 + i] >> 32);
  }
  hipLaunchKernelGGL(g1_multiples_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, b, first, (uint64_t)count, out_dev);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int plk_dev_coset_ntt(plk_ctx* ctx, void* stream, const uint64_t* const* in, size_t batch, size_t n_in, size_t N, const uint64_t shift[4],
                      int inverse, uint64_t* const* out) {
  STREAM_CALL(ctx, stream);
  int rc;
  if (!in || !out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  for (size_t done = 0; done < batch; done += NTT_MAX_BATCH) {
    const size_t m = std::min<size_t>(NTT_MAX_BATCH, batch - done);
    if ((rc = launch_coset_ntt(ctx, (hipStream_t)stream, in + done, m, n_in, N, shift, inverse != 0, out + done))) return rc;
  }
  return PLK_OK;
}

int plk_dev_quotient(plk_ctx* ctx, void* stream, const uint64_t* const* numerators, size_t count, const uint64_t alpha[4], size_t n, size_t N,
                     const uint64_t shift[4], uint64_t* coeffs_out) {
  STREAM_CALL(ctx, stream);
  return launch_quotient(ctx, (hipStream_t)stream, numerators, count, alpha, n, N, shift, coeffs_out);
}

int plk_dev_gate_table_fill(plk_ctx* ctx, void* stream, const uint8_t* row_ops, const uint64_t* const node_ids[4], size_t n,
                            const uint64_t* node_values, size_t n_values, const uint64_t default_val[4],
                            uint64_t* const wires_out[4], uint64_t* const sel_out[9]) {
  STREAM_CALL(ctx, stream);
  if (!node_ids || !wires_out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_gate_table_fill(ctx, (hipStream_t)stream, row_ops, node_ids, n, node_values, n_values, default_val, wires_out, sel_out);
}

int plk_dev_sigma_table(plk_ctx* ctx, void* stream, const uint64_t* const node_ids[4], size_t n, uint64_t* const sigma_out[4]) {
  STREAM_CALL(ctx, stream);
  if (!node_ids || !sigma_out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_sigma_table(ctx, (hipStream_t)stream, node_ids, n, sigma_out);
}

int plk_dev_gate_constraint_eval(plk_ctx* ctx, void* stream, const uint64_t* const wire_evals[4],
                                 const uint64_t* const sel_evals[9], size_t N, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (!wire_evals) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_gate_eval(ctx, (hipStream_t)stream, wire_evals, sel_evals, N, out);
}

int plk_dev_perm_constraint_eval(plk_ctx* ctx, void* stream, const uint64_t* z_evals, const uint64_t* const sigma_evals[4],
                                 const uint64_t* const wire_evals[4], size_t N, const uint64_t beta[4],
                                 const uint64_t gamma[4], size_t z_shift, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (!sigma_evals || !wire_evals) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_perm_eval(ctx, (hipStream_t)stream, z_evals, sigma_evals, wire_evals, N, beta, gamma, z_shift, out);
}

int plk_dev_perm_constraint_eval_range(plk_ctx* ctx, void* stream, const uint64_t* z_evals, const uint64_t* const sigma_evals[4],
                                       const uint64_t* const wire_evals[4], size_t count, size_t index0, const uint64_t beta[4],
                                       const uint64_t gamma[4], size_t z_shift, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (!sigma_evals || !wire_evals) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_perm_eval(ctx, (hipStream_t)stream, z_evals, sigma_evals, wire_evals, count, beta, gamma, z_shift, out, index0, true);
}

int plk_dev_round23(plk_ctx* ctx, void* stream, const uint64_t* const wires[4], const uint64_t* const sel[9],
                    const uint64_t* const sigma[4], size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4],
                    size_t z_shift, uint64_t* z_out, uint64_t* gate_out, uint64_t* perm_out) {
  STREAM_CALL(ctx, stream);
  if (!wires || !sigma) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  return launch_round23(ctx, (hipStream_t)stream, wires, sel, sigma, n, N, beta, gamma, z_shift, z_out, gate_out, perm_out);
}

int plk_dev_extend4(plk_ctx* ctx, void* stream, const uint64_t* const* cols, const uint32_t* kinds, size_t batch, size_t n,
                    uint64_t* const* coef_out, uint64_t* const* eval_out, size_t chunk, size_t chunk_stride, size_t halo) {
  STREAM_CALL(ctx, stream);
  int rc;
  if (!cols || !eval_out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  for (size_t done = 0; done < batch; done += NTT_MAX_BATCH) {
    const size_t m = std::min<size_t>(NTT_MAX_BATCH, batch - done);
    if ((rc = launch_extend4(ctx, (hipStream_t)stream, cols + done, kinds ? kinds + done : nullptr, m, n, coef_out ? coef_out + done : nullptr,
                             eval_out + done, chunk, chunk_stride, halo)))
      return rc;
  }
  return PLK_OK;
}

int plk_dev_ntt_strided(plk_ctx* ctx, void* stream, const uint64_t* in, size_t in_stride, uint32_t kind, size_t count,
                        size_t n_in, size_t N, int inverse, uint64_t* out, size_t out_stride) {
  STREAM_CALL(ctx, stream);
  int rc;
  if (count == 0) return PLK_OK;
  if (!out || (!in && n_in)) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (in_stride < n_in || out_stride < N) return fail(ctx, PLK_ERR_ARG, "stride shorter than a column");
  const uint64_t* ins[1] = {in};
  uint64_t* outs[1] = {out};
  const uint32_t kinds[1] = {kind};
  for (size_t done = 0; done < count; done += 32768) {   // grid.y limit
    const size_t m = std::min<size_t>(32768, count - done);
    ins[0] = in + done * in_stride * 4;
    outs[0] = out + done * out_stride * 4;
    if ((rc = launch_ntt(ctx, (hipStream_t)stream, ins, kinds, m, n_in, N, inverse != 0, outs, in_stride, out_stride))) return rc;
  }
  return PLK_OK;
}

int plk_dev_fourstep_cols(plk_ctx* ctx, void* stream, const uint64_t* recv1, size_t N, size_t ranks, size_t rank, int inverse, uint64_t* send2_out) {
  STREAM_CALL(ctx, stream);
  return launch_fs_cols(ctx, (hipStream_t)stream, recv1, N, ranks, rank, inverse != 0, send2_out);
}

int plk_dev_fourstep_rows(plk_ctx* ctx, void* stream, const uint64_t* recv2, size_t N, size_t ranks, size_t rank, int inverse, int natural_layout,
                          uint64_t* out) {
  STREAM_CALL(ctx, stream);
  return launch_fs_rows(ctx, (hipStream_t)stream, recv2, N, ranks, rank, inverse != 0, natural_layout != 0, out);
}

int plk_dev_mul(plk_ctx* ctx, void* stream, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (n == 0) return PLK_OK;
  if (!a || !b || !out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     a, b, (uint64_t)n, out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int plk_dev_twiddle_block(plk_ctx* ctx, void* stream, size_t N, int inverse, int scale_by_inverse_n, size_t rows, size_t cols,
                          size_t row0, size_t col0, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (rows * cols == 0) return PLK_OK;
  if (!out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Fe w = root_of_unity(log_n);
  if (inverse) w = fe_inverse(w);
  const Fe mul = scale_by_inverse_n ? fe_inverse(fe_from_u64((uint64_t)N)) : fe_one();
  const uint64_t total = (uint64_t)rows * cols;
  hipLaunchKernelGGL(twiddle_block_kernel, dim3((unsigned)((total + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, out, (uint64_t)rows, (uint64_t)cols, (uint64_t)row0, (uint64_t)col0, w, mul);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int plk_dev_status(plk_ctx* ctx, void* stream) {
  STREAM_CALL(ctx, stream);
  return fetch_status(ctx, (hipStream_t)stream);
}

int plk_dev_to_montgomery(plk_ctx* ctx, void* stream, const uint64_t* in, size_t n, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (n == 0) return PLK_OK;
  if (!in || !out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, in, (uint64_t)n, fe_r2(), out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

int plk_dev_from_montgomery(plk_ctx* ctx, void* stream, const uint64_t* in, size_t n, uint64_t* out) {
  STREAM_CALL(ctx, stream);
  if (n == 0) return PLK_OK;
  if (!in || !out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Fe raw_one = fe_zero();
  raw_one.l[0] = 1;
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + EW_THREADS - 1) / EW_THREADS)), dim3(EW_THREADS), 0,
                     (hipStream_t)stream, in, (uint64_t)n, raw_one, out);
  HIP_TRY(ctx, hipGetLastError());
  return PLK_OK;
}

// ---------------------------------------------------------------------------- host API

int plk_intt(plk_ctx* ctx, const uint64_t* evals, size_t evals_len, size_t domain_size, uint64_t* coeffs_out) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, domain_size, &log_n))) return rc;
  if (evals_len != domain_size) return fail(ctx, PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
  if (!evals || !coeffs_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  void* d;
  if ((rc = st.upload(evals, domain_size * ELEM, &d))) return rc;
  const uint64_t* in[1] = {(const uint64_t*)d};
  uint64_t* out[1] = {(uint64_t*)d};
  if ((rc = launch_ntt(ctx, ctx->stream, in, nullptr, 1, domain_size, domain_size, true, out))) return rc;
  { int rc_ = download(ctx, coeffs_out, d, domain_size * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_build_and_convert(plk_ctx* ctx, size_t domain_size, plk_builder_fn builder, void* user, uint64_t* coeffs_out) {
  HOST_CALL(ctx);
  int rc, log_n;
  if (!builder || !coeffs_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  // the domain is checked before anything is sized by it: a bogus size must come back as a code, not as an
  // exception leaving an extern "C" function
  if ((rc = check_domain(ctx, domain_size, &log_n))) return rc;
  uint64_t* evals = new (std::nothrow) uint64_t[domain_size * 4 + 4];
  if (!evals) return fail(ctx, PLK_ERR_NOMEM, "no host memory for " + std::to_string(domain_size) + " evaluations");
  const size_t produced = builder(user, evals, domain_size);
  if (produced != domain_size) rc = fail(ctx, PLK_ERR_LENGTH, "Domain size and builder evaluations length should match");
  else rc = plk_intt(ctx, evals, produced, domain_size, coeffs_out);
  delete[] evals;
  return rc;
}

int plk_ntt(plk_ctx* ctx, const uint64_t* coeffs, size_t n_coeffs, size_t domain_size, uint64_t* evals_out) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, domain_size, &log_n))) return rc;
  if (n_coeffs > domain_size) return fail(ctx, PLK_ERR_LENGTH, "more coefficients than domain points");
  if ((!coeffs && n_coeffs) || !evals_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  void *d_in, *d_out;
  if ((rc = st.upload(coeffs, n_coeffs * ELEM, &d_in))) return rc;
  if ((rc = st.alloc(domain_size * ELEM, &d_out))) return rc;
  const uint64_t* in[1] = {(const uint64_t*)d_in};
  uint64_t* out[1] = {(uint64_t*)d_out};
  if ((rc = launch_ntt(ctx, ctx->stream, in, nullptr, 1, n_coeffs, domain_size, false, out))) return rc;
  { int rc_ = download(ctx, evals_out, d_out, domain_size * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_gate_polys(plk_ctx* ctx, const uint64_t* const wires[4], const uint64_t* const sel[9], size_t n, size_t domain_size,
                   uint64_t* const wires_out[4], uint64_t* const sel_out[9]) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, domain_size, &log_n))) return rc;
  if (n != domain_size) return fail(ctx, PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
  if (!wires || !wires_out) return fail(ctx, PLK_ERR_ARG, "null pointer table");
  Staged st(ctx);
  const uint64_t* in[13]; uint64_t* out[13]; uint64_t* host_out[13]; size_t m = 0;
  for (int k = 0; k < 4; ++k) {
    if (!wires[k] || !wires_out[k]) return fail(ctx, PLK_ERR_ARG, "null wire column");
    void* d; if ((rc = st.upload(wires[k], n * ELEM, &d))) return rc;
    in[m] = (const uint64_t*)d; out[m] = (uint64_t*)d; host_out[m] = wires_out[k]; ++m;
  }
  for (int k = 0; k < 9; ++k) {
    if (!sel || !sel[k]) continue;
    if (!sel_out || !sel_out[k]) return fail(ctx, PLK_ERR_ARG, "selector present but no output column");
    void* d; if ((rc = st.upload(sel[k], n * ELEM, &d))) return rc;
    in[m] = (const uint64_t*)d; out[m] = (uint64_t*)d; host_out[m] = sel_out[k]; ++m;
  }
  if ((rc = launch_ntt(ctx, ctx->stream, in, nullptr, m, n, domain_size, true, out))) return rc;
  for (size_t b = 0; b < m; ++b) { int rc_ = download(ctx, host_out[b], out[b], n * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_gate_constraints(plk_ctx* ctx, const uint64_t* const wire_coeffs[4], const uint64_t* const sel_coeffs[9], size_t n_coeffs,
                         size_t N, uint64_t* out_host) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (n_coeffs > N) return fail(ctx, PLK_ERR_LENGTH, "more coefficients than domain points");
  if (!wire_coeffs || !out_host) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  const uint64_t* in[13]; uint64_t* ev[13]; int col[13]; size_t m = 0;
  for (int k = 0; k < 4; ++k) {
    if (!wire_coeffs[k] && n_coeffs) return fail(ctx, PLK_ERR_ARG, "null wire column");
    void *d, *e; if ((rc = st.upload(wire_coeffs[k], n_coeffs * ELEM, &d))) return rc; if ((rc = st.alloc(N * ELEM, &e))) return rc;
    in[m] = (const uint64_t*)d; ev[m] = (uint64_t*)e; col[m] = k; ++m;
  }
  for (int k = 0; k < 9; ++k) {
    if (!sel_coeffs || !sel_coeffs[k]) continue;
    void *d, *e; if ((rc = st.upload(sel_coeffs[k], n_coeffs * ELEM, &d))) return rc; if ((rc = st.alloc(N * ELEM, &e))) return rc;
    in[m] = (const uint64_t*)d; ev[m] = (uint64_t*)e; col[m] = 4 + k; ++m;
  }
  if ((rc = launch_ntt(ctx, ctx->stream, in, nullptr, m, n_coeffs, N, false, ev))) return rc;
  const uint64_t* w[4]; const uint64_t* s[9] = {nullptr};
  for (size_t b = 0; b < m; ++b) { if (col[b] < 4) w[col[b]] = ev[b]; else s[col[b] - 4] = ev[b]; }
  void* d_out; if ((rc = st.alloc(N * ELEM, &d_out))) return rc;
  if ((rc = launch_gate_eval(ctx, ctx->stream, w, s, N, (uint64_t*)d_out))) return rc;
  { int rc_ = download(ctx, out_host, d_out, N * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_grand_product(plk_ctx* ctx, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n,
                      const uint64_t beta[4], const uint64_t gamma[4], uint64_t* z_out) {
  HOST_CALL(ctx);
  int rc;
  if (!wires || !sigma || !z_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  const uint64_t* w[4]; const uint64_t* sg[4];
  for (int k = 0; k < 4; ++k) {
    if (n && (!wires[k] || !sigma[k])) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(wires[k], n * ELEM, &d))) return rc; if ((rc = st.upload(sigma[k], n * 8, &e))) return rc;
    w[k] = (const uint64_t*)d; sg[k] = (const uint64_t*)e;
  }
  void* dz; if ((rc = st.alloc((n + 1) * ELEM, &dz))) return rc;
  if ((rc = launch_grand_product(ctx, ctx->stream, w, sg, n, beta, gamma, (uint64_t*)dz))) return rc;
  if ((rc = fetch_status(ctx, ctx->stream))) return rc;
  { int rc_ = download(ctx, z_out, dz, (n + 1) * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_msm_set_bases(plk_ctx* ctx, const uint64_t* bases_host, size_t n) {
  HOST_CALL(ctx);
  int rc;
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!bases_host && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (n == 0) { msm_free_table(ctx); return PLK_OK; }
  uint64_t* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**)&d, n * 96));
  hipError_t e = hipMemcpy(d, bases_host, n * 96, hipMemcpyHostToDevice);
  rc = e == hipSuccess ? msm_build_table(ctx, ctx->stream, d, n) : PLK_OK;
  (void)hipFree(d);
  HIP_TRY(ctx, e);
  return rc;
}

int plk_msm_g1(plk_ctx* ctx, const uint64_t* scalars_host, size_t n, uint64_t out_xy[12]) {
  HOST_CALL(ctx);
  int rc;
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!scalars_host && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  void* d;
  if ((rc = st.upload(scalars_host, n * ELEM, &d))) return rc;
  const uint64_t* one[1] = {(const uint64_t*)d};
  return launch_msm(ctx, ctx->stream, one, 1, n, out_xy);
}

int plk_g1_sum(const uint64_t* points_xy, size_t count, uint64_t out_xy[12]) {
  if ((!points_xy && count) || !out_xy) return PLK_ERR_ARG;
  G1Jac acc = g1_inf();
  for (size_t k = 0; k < count; ++k) {
    G1Affine p;
    for (int i = 0; i < 6; ++i) {
      p.x.l[2 * i] = (uint32_t)points_xy[12 * k + i]; p.x.l[2 * i + 1] = (uint32_t)(points_xy[12 * k + i] >> 32);
      p.y.l[2 * i] = (uint32_t)points_xy[12 * k + 6 + i]; p.y.l[2 * i + 1] = (uint32_t)(points_xy[12 * k + 6 + i] >> 32);
    }
    acc = g1_add_affine(acc, p);
  }
  const G1Affine r = g1_to_affine_host(acc);
  for (int i = 0; i < 6; ++i) {
    out_xy[i] = (uint64_t)r.x.l[2 * i] | ((uint64_t)r.x.l[2 * i + 1] << 32);
    out_xy[6 + i] = (uint64_t)r.y.l[2 * i] | ((uint64_t)r.y.l[2 * i + 1] << 32);
  }
  return PLK_OK;
}

int plk_msm_g1_batch(plk_ctx* ctx, const uint64_t* const scalars_host[], size_t count, size_t n, uint64_t* out_xy) {
  HOST_CALL(ctx);
  int rc;
  if (!Field::HAS_G1) return fail(ctx, PLK_ERR_ARG, std::string("commitments are sums in BLS12-381 G1: not available for ") + Field::NAME);
  if (!scalars_host && count) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  std::vector<const uint64_t*> d(count);
  for (size_t k = 0; k < count; ++k) {
    if (!scalars_host[k] && n) return fail(ctx, PLK_ERR_ARG, "null pointer");
    void* p;
    if ((rc = st.upload(scalars_host[k], n * ELEM, &p))) return rc;
    d[k] = (const uint64_t*)p;
  }
  return launch_msm(ctx, ctx->stream, d.data(), count, n, out_xy);
}

int plk_coset_ntt(plk_ctx* ctx, const uint64_t* in, size_t n_in, size_t domain_size, const uint64_t shift[4], int inverse, uint64_t* out) {
  HOST_CALL(ctx);
  int rc;
  if ((!in && n_in) || !out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  void *d_in, *d_out;
  if ((rc = st.upload(in, n_in * ELEM, &d_in))) return rc;
  if ((rc = st.alloc(domain_size * ELEM, &d_out))) return rc;
  const uint64_t* ip[1] = {(const uint64_t*)d_in};
  uint64_t* op[1] = {(uint64_t*)d_out};
  if ((rc = launch_coset_ntt(ctx, ctx->stream, ip, 1, n_in, domain_size, shift, inverse != 0, op))) return rc;
  { int rc_ = download(ctx, out, d_out, domain_size * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_quotient(plk_ctx* ctx, const uint64_t* const* numerators, size_t count, const uint64_t alpha[4], size_t n, size_t N,
                 const uint64_t shift[4], uint64_t* coeffs_out) {
  HOST_CALL(ctx);
  int rc;
  if (!numerators || !coeffs_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (count == 0 || count > QT_MAX_NUMERATORS) return fail(ctx, PLK_ERR_ARG, "between 1 and 8 numerators");
  Staged st(ctx);
  const uint64_t* d[QT_MAX_NUMERATORS];
  for (size_t i = 0; i < count; ++i) {
    if (!numerators[i]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void* p; if ((rc = st.upload(numerators[i], N * ELEM, &p))) return rc;
    d[i] = (const uint64_t*)p;
  }
  void* d_out;
  if ((rc = st.alloc(N * ELEM, &d_out))) return rc;
  if ((rc = launch_quotient(ctx, ctx->stream, d, count, alpha, n, N, shift, (uint64_t*)d_out))) return rc;
  { int rc_ = download(ctx, coeffs_out, d_out, N * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_gate_table_fill(plk_ctx* ctx, const uint8_t* row_ops, const uint64_t* const node_ids[4], size_t n,
                        const uint64_t* node_values, size_t n_values, const uint64_t default_val[4],
                        uint64_t* const wires_out[4], uint64_t* const sel_out[9]) {
  HOST_CALL(ctx);
  int rc;
  if (!node_ids || !wires_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (n == 0) return PLK_OK;
  if (!row_ops) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  const uint64_t* nd[4]; uint64_t* w[4]; uint64_t* sl[9];
  void *ops, *vals;
  if ((rc = st.upload(row_ops, n, &ops))) return rc;
  if ((rc = st.upload(node_values, n_values * ELEM, &vals))) return rc;
  for (int k = 0; k < 4; ++k) {
    if (!node_ids[k] || !wires_out[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(node_ids[k], n * 8, &d))) return rc; if ((rc = st.alloc(n * ELEM, &e))) return rc;
    nd[k] = (const uint64_t*)d; w[k] = (uint64_t*)e;
  }
  for (int k = 0; k < 9; ++k) {
    sl[k] = nullptr;
    if (sel_out && sel_out[k]) { void* e; if ((rc = st.alloc(n * ELEM, &e))) return rc; sl[k] = (uint64_t*)e; }
  }
  if ((rc = launch_gate_table_fill(ctx, ctx->stream, (const uint8_t*)ops, nd, n, (const uint64_t*)vals, n_values, default_val, w, sl))) return rc;
  for (int k = 0; k < 4; ++k) { int rc_ = download(ctx, wires_out[k], w[k], n * ELEM); if (rc_) return rc_; }
  for (int k = 0; k < 9; ++k) if (sl[k]) { int rc_ = download(ctx, sel_out[k], sl[k], n * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_sigma_table(plk_ctx* ctx, const uint64_t* const node_ids[4], size_t n, uint64_t* const sigma_out[4]) {
  HOST_CALL(ctx);
  int rc;
  if (!node_ids || !sigma_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  if (n == 0) return PLK_OK;
  Staged st(ctx);
  const uint64_t* nd[4]; uint64_t* sg[4];
  for (int k = 0; k < 4; ++k) {
    if (!node_ids[k] || !sigma_out[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(node_ids[k], n * 8, &d))) return rc; if ((rc = st.alloc(n * 8, &e))) return rc;
    nd[k] = (const uint64_t*)d; sg[k] = (uint64_t*)e;
  }
  if ((rc = launch_sigma_table(ctx, ctx->stream, nd, n, sg))) return rc;
  for (int k = 0; k < 4; ++k) { int rc_ = download(ctx, sigma_out[k], sg[k], n * 8); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_perm_polys(plk_ctx* ctx, const uint64_t* const wires[4], const uint64_t* const sigma[4], size_t n, size_t domain_size,
                   const uint64_t beta[4], const uint64_t gamma[4], uint64_t* const sigma_coeffs_out[4], uint64_t* z_coeffs_out,
                   uint64_t* z_last_out) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, domain_size, &log_n))) return rc;
  if (n != domain_size) return fail(ctx, PLK_ERR_LENGTH, plk_strerror(PLK_ERR_LENGTH));
  if (!wires || !sigma || !sigma_coeffs_out || !z_coeffs_out) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  const uint64_t* w[4]; const uint64_t* sg[4];
  for (int k = 0; k < 4; ++k) {
    if (!wires[k] || !sigma[k] || !sigma_coeffs_out[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(wires[k], n * ELEM, &d))) return rc; if ((rc = st.upload(sigma[k], n * 8, &e))) return rc;
    w[k] = (const uint64_t*)d; sg[k] = (const uint64_t*)e;
  }
  void* dz; if ((rc = st.alloc((n + 1) * ELEM, &dz))) return rc;
  if ((rc = launch_grand_product(ctx, ctx->stream, w, sg, n, beta, gamma, (uint64_t*)dz))) return rc;
  if ((rc = fetch_status(ctx, ctx->stream))) return rc;
  if (z_last_out) HIP_TRY(ctx, hipMemcpyAsync(z_last_out, (uint64_t*)dz + 4 * n, ELEM, hipMemcpyDeviceToHost, ctx->stream));
  const uint64_t* in[5]; uint64_t* out[5]; uint32_t kinds[5];
  for (int k = 0; k < 4; ++k) {
    void* c; if ((rc = st.alloc(n * ELEM, &c))) return rc;
    in[k] = sg[k]; kinds[k] = NTT_IN_U64; out[k] = (uint64_t*)c;
  }
  in[4] = (const uint64_t*)dz; kinds[4] = NTT_IN_FIELD; out[4] = (uint64_t*)dz;
  if ((rc = launch_ntt(ctx, ctx->stream, in, kinds, 5, n, domain_size, true, out))) return rc;
  for (int k = 0; k < 4; ++k) { int rc_ = download(ctx, sigma_coeffs_out[k], out[k], n * ELEM); if (rc_) return rc_; }
  { int rc_ = download(ctx, z_coeffs_out, dz, n * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_perm_constraints(plk_ctx* ctx, const uint64_t* z_coeffs, const uint64_t* const sigma_coeffs[4],
                         const uint64_t* const wire_coeffs[4], size_t n_coeffs, size_t N, const uint64_t beta[4],
                         const uint64_t gamma[4], size_t z_shift, uint64_t* out_host) {
  HOST_CALL(ctx);
  int rc, log_n;
  if ((rc = check_domain(ctx, N, &log_n))) return rc;
  if (n_coeffs > N) return fail(ctx, PLK_ERR_LENGTH, "more coefficients than domain points");
  if (!sigma_coeffs || !wire_coeffs || !out_host || (!z_coeffs && n_coeffs)) return fail(ctx, PLK_ERR_ARG, "null pointer");
  Staged st(ctx);
  const uint64_t* in[9]; uint64_t* ev[9];
  const uint64_t* host_cols[9] = {z_coeffs, sigma_coeffs[0], sigma_coeffs[1], sigma_coeffs[2], sigma_coeffs[3],
                                  wire_coeffs[0], wire_coeffs[1], wire_coeffs[2], wire_coeffs[3]};
  for (int b = 0; b < 9; ++b) {
    if (!host_cols[b] && n_coeffs) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(host_cols[b], n_coeffs * ELEM, &d))) return rc; if ((rc = st.alloc(N * ELEM, &e))) return rc;
    in[b] = (const uint64_t*)d; ev[b] = (uint64_t*)e;
  }
  if ((rc = launch_ntt(ctx, ctx->stream, in, nullptr, 9, n_coeffs, N, false, ev))) return rc;
  const uint64_t* sg[4] = {ev[1], ev[2], ev[3], ev[4]};
  const uint64_t* w[4] = {ev[5], ev[6], ev[7], ev[8]};
  void* d_out; if ((rc = st.alloc(N * ELEM, &d_out))) return rc;
  if ((rc = launch_perm_eval(ctx, ctx->stream, ev[0], sg, w, N, beta, gamma, z_shift, (uint64_t*)d_out))) return rc;
  { int rc_ = download(ctx, out_host, d_out, N * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

int plk_round23(plk_ctx* ctx, const uint64_t* const wires[4], const uint64_t* const sel[9], const uint64_t* const sigma[4], size_t n,
                size_t N, const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* z_out, uint64_t* gate_out,
                uint64_t* perm_out) {
  HOST_CALL(ctx);
  int rc, l1, l2;
  if ((rc = check_domain(ctx, n, &l1))) return rc;
  if ((rc = check_domain(ctx, N, &l2))) return rc;
  if (!wires || !sigma || !beta || !gamma) return fail(ctx, PLK_ERR_ARG, "null pointer");
  for (int k = 0; k < 4; ++k) if (!wires[k] || !sigma[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
  if (!fe_is_canonical(fe_from_limbs64(beta)) || !fe_is_canonical(fe_from_limbs64(gamma)))   // before anything is uploaded
    return fail(ctx, PLK_ERR_ARG, "beta/gamma not reduced mod p");
  if (n <= N && r23_coset_form(n, N, l1))
    return round23_host_pipelined(ctx, wires, sel, sigma, n, N, beta, gamma, z_shift, z_out, gate_out, perm_out);
  Staged st(ctx);
  const uint64_t* w[4]; const uint64_t* sg[4]; const uint64_t* s[9] = {nullptr};
  for (int k = 0; k < 4; ++k) {
    if (!wires[k] || !sigma[k]) return fail(ctx, PLK_ERR_ARG, "null column pointer");
    void *d, *e; if ((rc = st.upload(wires[k], n * ELEM, &d))) return rc; if ((rc = st.upload(sigma[k], n * 8, &e))) return rc;
    w[k] = (const uint64_t*)d; sg[k] = (const uint64_t*)e;
  }
  for (int k = 0; k < 9; ++k) {
    if (!sel || !sel[k]) continue;
    void* d; if ((rc = st.upload(sel[k], n * ELEM, &d))) return rc;
    s[k] = (const uint64_t*)d;
  }
  void *dz, *dg, *dp;
  if ((rc = st.alloc((n + 1) * ELEM, &dz))) return rc;
  if ((rc = st.alloc(N * ELEM, &dg))) return rc;
  if ((rc = st.alloc(N * ELEM, &dp))) return rc;
  if ((rc = launch_round23(ctx, ctx->stream, w, s, sg, n, N, beta, gamma, z_shift, (uint64_t*)dz, (uint64_t*)dg, (uint64_t*)dp))) return rc;
  if ((rc = fetch_status(ctx, ctx->stream))) return rc;
  if (z_out) { int rc_ = download(ctx, z_out, dz, (n + 1) * ELEM); if (rc_) return rc_; }
  if (gate_out) { int rc_ = download(ctx, gate_out, dg, N * ELEM); if (rc_) return rc_; }
  if (perm_out) { int rc_ = download(ctx, perm_out, dp, N * ELEM); if (rc_) return rc_; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return PLK_OK;
}

}  // extern "C"
