// G ranks in ONE process, one GPU each (declared in include/plonk_hip.h: plk_node_*).
//
// Host code only.  Everything a rank computes goes through the library's public device entry points
// (plk_dev_fourstep_cols / _rows, plk_dev_round23, plk_dev_grand_product, plk_dev_extend4, the two constraint
// evaluations, plk_dev_status) on that rank's own context; this file adds what turns G such contexts into one node: a
// host thread and a stream per rank, the three all-to-all exchanges of the four-step transform, the one exchange of
// a proof whose columns are dealt to the ranks, and the bookkeeping that lets the exchanges run with no host
// synchronisation in between.
//
// Three modes: plk_node_ntt (one transform sharded over the ranks), plk_node_round23 (one proof: columns dealt to the
// ranks, then every rank evaluates its range of points), plk_node_round23_batch (whole proofs dealt to the ranks, no
// exchange); the _host forms take host arrays.
//
// The reference has no counterpart: ark-poly's transform is one shared-memory job (src/polynomial_utils.rs:29,
// src/execution_trace.rs:362-370) and nothing in the reference is distributed.
//
// Exchange, two forms chosen at plk_node_create:
//   PLK_NODE_RCCL  ncclAllToAll on communicators from ncclCommInitAll, enqueued on the ranks' streams.  One message
//                  per peer: on an 8 x MI355X node one per xGMI link (N = 2^24: 7 messages of 8 MiB per rank and
//                  exchange).  RCCL is bound with dlopen when the first such node is created.
//   PLK_NODE_COPY  the sender pushes every block straight to its place in the receiver's buffer with asynchronous
//                  device-to-device copies on its own stream (the 2-D copy does the pack of exchange 1 and the
//                  transposition of exchange 3 on the way) and records an event; the receiver's stream waits for the
//                  G events.  Between GPUs these are peer copies; with several ranks on ONE device they are plain
//                  copies, which is how G = 2, 4, 8 are rehearsed on a single GPU.
//
// Buffers of rank g (N / G elements each, grow-only): A = receive buffer of exchange 1, later the send buffer of
// exchange 3; B = send buffer of exchange 2 (and pack / unpack area of the RCCL form); C = receive buffer of exchange 2.
// plk_node_round23 has a send and a receive buffer of its own (F, G), the _host forms their staging buffers (D, E, H).
// A call leaves an event `done` on every rank's stream; the next call's streams wait for all of them before they
// touch another rank's buffers.
#include "../../include/plonk_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: the functions are bound at run time

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t ELEM = 32;

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
  bool load() {
    if (handle) return true;
    // the soname: a process that already holds a copy (PyTorch ships one under torch/lib) gets that one back
    handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!handle) { error = std::string("dlopen(librccl.so.1): ") + dlerror(); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
    AllToAll = (decltype(AllToAll))dlsym(handle, "ncclAllToAll");
    GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllToAll || !GetErrorString) {
      error = "librccl.so.1 lacks ncclCommInitAll / ncclCommDestroy / ncclAllToAll / ncclGetErrorString";
      handle = nullptr;
      return false;
    }
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// G persistent host threads; run(f) executes f(g) on thread g for every g and returns when all are done
struct Workers {
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv_start, cv_done;
  std::function<void(size_t)> job;
  uint64_t generation = 0;
  size_t pending = 0;
  bool stop = false;
  void start(size_t G) {
    for (size_t g = 0; g < G; ++g)
      threads.emplace_back([this, g] {
        uint64_t seen = 0;
        for (;;) {
          std::function<void(size_t)> j;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_start.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            j = job;
          }
          j(g);
          std::lock_guard<std::mutex> lk(m);
          if (--pending == 0) cv_done.notify_one();
        }
      });
  }
  void run(const std::function<void(size_t)>& f) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = f; pending = threads.size(); ++generation;
    }
    cv_start.notify_all();
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

// every rank's thread arrives, then all leave (an event must have been recorded before another thread makes a stream
// wait for it: a wait on an event never recorded is no wait at all)
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  size_t count = 1, waiting = 0;
  uint64_t phase = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const uint64_t p = phase;
    if (++waiting == count) { waiting = 0; ++phase; cv.notify_all(); }
    else cv.wait(lk, [&] { return phase != p; });
  }
};

struct Rank {
  int device = 0;
  plk_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;           // the context's own stream
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};   // "my blocks of exchange k have been pushed"
  hipEvent_t done = nullptr;              // "my part of the last call has finished and my output is complete"
  // A, B, C; D, E = shards of plk_node_ntt_host; F, G = send and receive buffers of plk_node_round23; H = the columns
  // and outputs of plk_node_round23_host
  void* buf[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  ncclComm_t comm = nullptr;
  int rc = PLK_OK;
  std::string err;
};

}  // namespace

struct plk_node {
  std::vector<Rank> r;
  int exchange = PLK_NODE_COPY;
  Workers workers;
  Barrier barrier;
  bool have_done = false;
  std::string last_error;
};

namespace {

int node_fail(plk_node* node, int code, const std::string& msg) {
  if (node) node->last_error = msg;
  return code;
}

bool hip_ok(Rank& me, hipError_t e, const char* what) {
  if (e == hipSuccess) return true;
  if (me.rc == PLK_OK) {
    me.rc = e == hipErrorOutOfMemory ? PLK_ERR_NOMEM : PLK_ERR_HIP;
    me.err = std::string(what) + ": " + hipGetErrorString(e);
  }
  (void)hipGetLastError();
  return false;
}
bool plk_ok(Rank& me, int rc, const char* what) {
  if (rc == PLK_OK) return true;
  if (me.rc == PLK_OK) { me.rc = rc; me.err = std::string(what) + ": " + plk_last_error(me.ctx); }
  return false;
}
bool nccl_ok(Rank& me, ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return true;
  if (me.rc == PLK_OK) { me.rc = PLK_ERR_HIP; me.err = std::string(what) + ": " + g_rccl.GetErrorString(e); }
  return false;
}
#define HIP_OK(expr) hip_ok(me, (expr), #expr)
#define PLK_OKAY(expr) plk_ok(me, (expr), #expr)
#define NCCL_OK(expr) nccl_ok(me, (expr), #expr)

// Buffers [first, last] of every rank hold at least `bytes`.  Runs on the calling thread before the ranks' threads
// start: other ranks write into a rank's buffers, so their addresses must be settled before anyone reads them, and a
// buffer is only replaced once nothing enqueued anywhere on the node can still touch it.
int ensure_buffers(plk_node* node, int first, int last, size_t bytes) {
  bool need = false;
  for (auto& me : node->r)
    for (int b = first; b <= last; ++b) need = need || me.cap[b] < bytes;
  if (!need) return PLK_OK;
  int rc = plk_node_sync(node);
  if (rc) return rc;
  for (auto& me : node->r)
    for (int b = first; b <= last; ++b) {
      if (me.cap[b] >= bytes) continue;
      if (me.buf[b]) { plk_dev_free(me.ctx, me.buf[b]); me.buf[b] = nullptr; me.cap[b] = 0; }
      if ((rc = plk_dev_alloc(me.ctx, bytes, &me.buf[b]))) return node_fail(node, rc, std::string("plk_dev_alloc: ") + plk_last_error(me.ctx));
      me.cap[b] = bytes;
    }
  return PLK_OK;
}

// collects the ranks' results of one job: the first failing rank's code and text
int collect(plk_node* node, const char* what) {
  for (size_t g = 0; g < node->r.size(); ++g)
    if (node->r[g].rc != PLK_OK)
      return node_fail(node, node->r[g].rc, std::string(what) + ", rank " + std::to_string(g) + ": " + node->r[g].err);
  return PLK_OK;
}

int log2_exact(size_t n) {
  if (n == 0 || (n & (n - 1))) return -1;
  int l = 0;
  while ((size_t(1) << l) < n) ++l;
  return l;
}

// One rank's part of the transform; every rank's thread runs this at the same time.  Whatever fails, every barrier
// is still reached.
void ntt_rank(plk_node* node, size_t g, const uint64_t* const* in, size_t N, size_t N1, size_t N2, bool inverse, bool natural,
              uint64_t* const* out) {
  const size_t G = node->r.size();
  Rank& me = node->r[g];
  const size_t r1 = N1 / G, c2 = N2 / G, blk = r1 * c2;          // elements per (source, destination) pair
  const size_t shard = (N / G) * ELEM;
  const int inv = inverse ? 1 : 0;
  me.rc = PLK_OK; me.err.clear();
  bool ok = HIP_OK(hipSetDevice(me.device));
  (void)shard;
  char *A = (char*)me.buf[0], *B = (char*)me.buf[1], *C = (char*)me.buf[2];
  hipStream_t s = me.stream;
  // the previous call may still be reading or writing the buffers this one is about to write on other ranks
  if (ok && node->have_done)
    for (size_t h = 0; h < G && ok; ++h) ok = HIP_OK(hipStreamWaitEvent(s, node->r[h].done, 0));
  auto wait_all = [&](int k) {
    for (size_t h = 0; h < G && ok; ++h)
      if (h != g) ok = HIP_OK(hipStreamWaitEvent(s, node->r[h].ev[k], 0));
  };

  if (node->exchange == PLK_NODE_RCCL) {
    // exchange 1: pack my rows' columns by destination, [r1][G][c2] -> B = [h][r1][c2], then all-to-all into A
    for (size_t h = 0; h < G && ok; ++h)
      ok = HIP_OK(hipMemcpy2DAsync(B + h * blk * ELEM, c2 * ELEM, (const char*)in[g] + h * c2 * ELEM, N2 * ELEM, c2 * ELEM, r1,
                                   hipMemcpyDeviceToDevice, s));
    if (ok) ok = NCCL_OK(g_rccl.AllToAll(B, A, blk * 4, ncclUint64, me.comm, s));
    if (ok) ok = PLK_OKAY(plk_dev_fourstep_cols(me.ctx, s, (const uint64_t*)A, N, G, g, inv, (uint64_t*)B));
    if (ok) ok = NCCL_OK(g_rccl.AllToAll(B, C, blk * 4, ncclUint64, me.comm, s));       // [source][k1 local][c2]
    if (!natural) {
      if (ok) ok = PLK_OKAY(plk_dev_fourstep_rows(me.ctx, s, (const uint64_t*)C, N, G, g, inv, 0, out[g]));
    } else {
      if (ok) ok = PLK_OKAY(plk_dev_fourstep_rows(me.ctx, s, (const uint64_t*)C, N, G, g, inv, 1, (uint64_t*)A));
      if (ok) ok = NCCL_OK(g_rccl.AllToAll(A, B, blk * 4, ncclUint64, me.comm, s));     // [source][k2 local][k1 local]
      // my range of X is [k2 local][source][k1 local]: one 2-D copy per source
      for (size_t h = 0; h < G && ok; ++h)
        ok = HIP_OK(hipMemcpy2DAsync((char*)out[g] + h * r1 * ELEM, N1 * ELEM, B + h * blk * ELEM, r1 * ELEM, r1 * ELEM, c2,
                                     hipMemcpyDeviceToDevice, s));
    }
    if (ok) ok = HIP_OK(hipEventRecord(me.done, s));
    return;
  }

  // PLK_NODE_COPY
  // exchange 1: columns [h c2, (h + 1) c2) of my r1 rows -> block g of rank h's A (its matrix [N1][c2] in row order)
  for (size_t h = 0; h < G && ok; ++h)
    ok = HIP_OK(hipMemcpy2DAsync((char*)node->r[h].buf[0] + g * blk * ELEM, c2 * ELEM, (const char*)in[g] + h * c2 * ELEM, N2 * ELEM,
                                 c2 * ELEM, r1, hipMemcpyDefault, s));
  if (ok) ok = HIP_OK(hipEventRecord(me.ev[0], s));
  node->barrier.wait();
  wait_all(0);
  if (ok) ok = PLK_OKAY(plk_dev_fourstep_cols(me.ctx, s, (const uint64_t*)A, N, G, g, inv, (uint64_t*)B));
  // exchange 2: rows [h r1, (h + 1) r1) of B = [k1][c2] -> block g of rank h's C
  for (size_t h = 0; h < G && ok; ++h)
    ok = HIP_OK(hipMemcpyAsync((char*)node->r[h].buf[2] + g * blk * ELEM, B + h * blk * ELEM, blk * ELEM, hipMemcpyDefault, s));
  if (ok) ok = HIP_OK(hipEventRecord(me.ev[1], s));
  node->barrier.wait();
  wait_all(1);
  if (!natural) {
    if (ok) ok = PLK_OKAY(plk_dev_fourstep_rows(me.ctx, s, (const uint64_t*)C, N, G, g, inv, 0, out[g]));
    if (ok) ok = HIP_OK(hipEventRecord(me.done, s));
    return;
  }
  // (A is free again: the pushes into it were waited for before the column phase, which has read it)
  if (ok) ok = PLK_OKAY(plk_dev_fourstep_rows(me.ctx, s, (const uint64_t*)C, N, G, g, inv, 1, (uint64_t*)A));
  // exchange 3: block h of A = [k2 local][k1 local] -> rank h's output at [k2 local][g][k1 local]
  for (size_t h = 0; h < G && ok; ++h)
    ok = HIP_OK(hipMemcpy2DAsync((char*)out[h] + g * r1 * ELEM, N1 * ELEM, A + h * blk * ELEM, r1 * ELEM, r1 * ELEM, c2, hipMemcpyDefault, s));
  if (ok) ok = HIP_OK(hipEventRecord(me.ev[2], s));
  node->barrier.wait();
  wait_all(2);
  if (ok) ok = HIP_OK(hipEventRecord(me.done, s));          // behind my own work and every push into my output
}

int ntt_check(plk_node* node, size_t N, size_t* N1, size_t* N2) {
  const int log_n = log2_exact(N);
  if (log_n < 0) return node_fail(node, PLK_ERR_NOT_POW2, "domain size " + std::to_string(N) + " is not a power of two");
  const size_t G = node->r.size();
  *N1 = size_t(1) << ((log_n + 1) / 2);
  *N2 = N / *N1;
  if (*N1 % G || *N2 % G || *N2 < G) return node_fail(node, PLK_ERR_ARG, std::to_string(G) + " ranks do not divide both sides of the " +
                                                                         std::to_string(*N1) + " x " + std::to_string(*N2) + " view of the domain");
  return PLK_OK;
}

}  // namespace

extern "C" {

int plk_node_create(const int* devices, size_t G, int exchange, plk_node** out) {
  return plk_node_create_field(devices, G, exchange, PLK_FIELD_BLS12_381_FR, out);
}

int plk_node_create_field(const int* devices, size_t G, int exchange, int field_id, plk_node** out) {
  if (!out) return PLK_ERR_ARG;
  *out = nullptr;
  if (field_id < 0 || field_id >= PLK_NUM_FIELDS) return PLK_ERR_ARG;
  if (!devices || G == 0 || G > 64 || (G & (G - 1)) || (exchange != PLK_NODE_RCCL && exchange != PLK_NODE_COPY)) return PLK_ERR_ARG;
  if (exchange == PLK_NODE_RCCL)
    for (size_t a = 0; a < G; ++a)
      for (size_t b = a + 1; b < G; ++b)
        if (devices[a] == devices[b]) {
          fprintf(stderr, "plk_node_create: RCCL wants one GPU per rank (device %d given twice); use PLK_NODE_COPY to put several ranks on one GPU\n", devices[a]);
          return PLK_ERR_ARG;
        }
  plk_node* node = new (std::nothrow) plk_node();
  if (!node) return PLK_ERR_NOMEM;
  node->exchange = exchange;
  node->r.resize(G);
  node->barrier.count = G;
  int rc = PLK_OK;
  for (size_t g = 0; g < G && rc == PLK_OK; ++g) {
    Rank& me = node->r[g];
    me.device = devices[g];
    rc = plk_ctx_create_field(me.device, field_id, &me.ctx);
    if (rc) break;
    me.stream = (hipStream_t)plk_ctx_stream(me.ctx);
    hipError_t e = hipSetDevice(me.device);
    for (auto& ev : me.ev) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&me.done, hipEventDisableTiming);
    if (e != hipSuccess) { fprintf(stderr, "plk_node_create: %s\n", hipGetErrorString(e)); rc = PLK_ERR_HIP; }
  }
  if (rc == PLK_OK && exchange == PLK_NODE_COPY) {
    // peer access between every pair of different GPUs (already enabled is fine)
    for (size_t a = 0; a < G; ++a)
      for (size_t b = 0; b < G; ++b) {
        if (node->r[a].device == node->r[b].device) continue;
        (void)hipSetDevice(node->r[a].device);
        const hipError_t e = hipDeviceEnablePeerAccess(node->r[b].device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
          fprintf(stderr, "plk_node_create: no peer access from GPU %d to GPU %d: %s\n", node->r[a].device, node->r[b].device, hipGetErrorString(e));
          rc = PLK_ERR_HIP;
        }
        (void)hipGetLastError();
      }
  }
  if (rc == PLK_OK && exchange == PLK_NODE_RCCL) {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (!g_rccl.load()) { fprintf(stderr, "plk_node_create: %s\n", g_rccl.error.c_str()); rc = PLK_ERR_HIP; }
    else {
      std::vector<ncclComm_t> comms(G, nullptr);
      const ncclResult_t e = g_rccl.CommInitAll(comms.data(), (int)G, devices);
      if (e != ncclSuccess) { fprintf(stderr, "plk_node_create: ncclCommInitAll: %s\n", g_rccl.GetErrorString(e)); rc = PLK_ERR_HIP; }
      else for (size_t g = 0; g < G; ++g) node->r[g].comm = comms[g];
    }
  }
  if (rc != PLK_OK) { plk_node_destroy(node); return rc; }
  node->workers.start(G);
  *out = node;
  return PLK_OK;
}

void plk_node_destroy(plk_node* node) {
  if (!node) return;
  node->workers.shutdown();
  for (auto& me : node->r) {
    if (!me.ctx) continue;
    (void)hipSetDevice(me.device);
    (void)hipDeviceSynchronize();
    if (me.comm) (void)g_rccl.CommDestroy(me.comm);
    for (auto& b : me.buf) if (b) plk_dev_free(me.ctx, b);
    for (auto& ev : me.ev) if (ev) (void)hipEventDestroy(ev);
    if (me.done) (void)hipEventDestroy(me.done);
    plk_ctx_destroy(me.ctx);
  }
  delete node;
}

size_t plk_node_ranks(const plk_node* node) { return node ? node->r.size() : 0; }
plk_ctx* plk_node_ctx(plk_node* node, size_t g) { return node && g < node->r.size() ? node->r[g].ctx : nullptr; }
const char* plk_node_last_error(const plk_node* node) { return node ? node->last_error.c_str() : ""; }

int plk_node_sync(plk_node* node) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  for (auto& me : node->r) {
    hipError_t e = hipSetDevice(me.device);
    if (e == hipSuccess) e = hipStreamSynchronize(me.stream);
    if (e != hipSuccess) return node_fail(node, PLK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  }
  return PLK_OK;
}

int plk_node_ntt(plk_node* node, const uint64_t* const* in_shards, size_t N, int inverse, int natural_output, uint64_t* const* out_shards) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  if (!in_shards || !out_shards) return node_fail(node, PLK_ERR_ARG, "null pointer table");
  size_t N1, N2;
  int rc = ntt_check(node, N, &N1, &N2);
  if (rc) return rc;
  for (size_t g = 0; g < node->r.size(); ++g)
    if (!in_shards[g] || !out_shards[g] || in_shards[g] == out_shards[g]) return node_fail(node, PLK_ERR_ARG, "distinct non-null shards expected");
  if ((rc = ensure_buffers(node, 0, 2, (N / node->r.size()) * ELEM))) return rc;
  node->workers.run([&](size_t g) { ntt_rank(node, g, in_shards, N, N1, N2, inverse != 0, natural_output != 0, out_shards); });
  node->have_done = true;      // (after a failure too: the events that were recorded are harmless to wait for)
  return collect(node, "plk_node_ntt");
}

int plk_node_ntt_host(plk_node* node, const uint64_t* in, size_t N, int inverse, uint64_t* out) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  if (!in || !out) return node_fail(node, PLK_ERR_ARG, "null pointer");
  size_t N1, N2;
  int rc = ntt_check(node, N, &N1, &N2);
  if (rc) return rc;
  const size_t G = node->r.size(), shard = (N / G) * ELEM;
  std::vector<const uint64_t*> ins(G);
  std::vector<uint64_t*> outs(G);
  if ((rc = ensure_buffers(node, 3, 4, shard))) return rc;
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    me.rc = PLK_OK; me.err.clear();
    if (!HIP_OK(hipSetDevice(me.device))) return;
    (void)PLK_OKAY(plk_dev_upload(me.ctx, me.stream, me.buf[3], (const char*)in + g * shard, shard));
  });
  if ((rc = collect(node, "plk_node_ntt_host (upload)"))) return rc;
  for (size_t g = 0; g < G; ++g) { ins[g] = (const uint64_t*)node->r[g].buf[3]; outs[g] = (uint64_t*)node->r[g].buf[4]; }
  if ((rc = plk_node_ntt(node, ins.data(), N, inverse, 1, outs.data()))) return rc;
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    me.rc = PLK_OK; me.err.clear();
    if (!HIP_OK(hipSetDevice(me.device))) return;
    (void)PLK_OKAY(plk_dev_download(me.ctx, me.stream, (char*)out + g * shard, me.buf[4], shard));   // behind `done` on this stream
  });
  return collect(node, "plk_node_ntt_host (download)");
}

int plk_node_round23_batch(plk_node* node, const plk_circuit_io* circuits, size_t count, size_t n, size_t N, size_t z_shift) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  if (!circuits && count) return node_fail(node, PLK_ERR_ARG, "null pointer");
  const size_t G = node->r.size();
  std::vector<long long> bad(G, -1);          // per rank: the first circuit that failed
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    me.rc = PLK_OK; me.err.clear();
    if (!HIP_OK(hipSetDevice(me.device))) { bad[g] = (long long)g; return; }
    auto enqueue = [&](size_t i) {
      const plk_circuit_io& c = circuits[i];
      return plk_dev_round23(me.ctx, me.stream, c.wires, c.sel, c.sigma, n, N, c.beta, c.gamma, z_shift, c.z_out, c.gate_out, c.perm_out);
    };
    // all of this rank's circuits back to back, one status check at the end ...
    int rc = PLK_OK;
    for (size_t i = g; i < count && rc == PLK_OK; i += G)
      if ((rc = enqueue(i))) { bad[g] = (long long)i; (void)plk_ok(me, rc, "plk_dev_round23"); }
    const int st = plk_dev_status(me.ctx, me.stream);
    if (rc != PLK_OK || st == PLK_OK) return;
    // ... and only if that reports a data-dependent failure, once more one by one to name the circuit
    for (size_t i = g; i < count; i += G) {
      int r2 = enqueue(i);
      if (r2 == PLK_OK) r2 = plk_dev_status(me.ctx, me.stream);
      if (r2 != PLK_OK) { bad[g] = (long long)i; (void)plk_ok(me, r2, "plk_dev_round23"); return; }
    }
    (void)plk_ok(me, st, "plk_dev_status");     // (did not show again)
    bad[g] = (long long)g;
  });
  long long first = -1;
  for (size_t g = 0; g < G; ++g)
    if (bad[g] >= 0 && (first < 0 || bad[g] < first)) first = bad[g];
  if (first < 0) return PLK_OK;
  const Rank& r = node->r[(size_t)first % G];
  return node_fail(node, r.rc, "plk_node_round23_batch, circuit " + std::to_string(first) + " (rank " + std::to_string((size_t)first % G) + "): " + r.err);
}

// ONE proof over the node's ranks (columns dealt to the ranks for interpolation and extension, one all-to-all, the
// constraints on N / G points per rank).  Column numbering: wires 0-3, selectors 4-12, sigma 13-16, Z 17; the columns
// present are dealt round-robin, rank g taking the ones at positions g, g + G, ... of that list.
int plk_node_round23(plk_node* node, const plk_circuit_io* per_rank, size_t n, size_t N, size_t z_shift) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  if (!per_rank) return node_fail(node, PLK_ERR_ARG, "null pointer");
  const size_t G = node->r.size();
  if (log2_exact(n) < 2 || N != 4 * n) return node_fail(node, PLK_ERR_ARG, "plk_node_round23 takes the extended domain N = 4 n, n a power of two >= 4");
  if (N % G) return node_fail(node, PLK_ERR_ARG, "the ranks must divide the evaluation domain");
  const size_t chunk = N / G;
  if (z_shift == 0 || z_shift > chunk) return node_fail(node, PLK_ERR_ARG, "z_shift must be between 1 and N / ranks");
  // the columns present (the same on every rank: taken from rank 0's table, checked against the others)
  int present[18]; size_t np = 0;
  for (int c = 0; c < 18; ++c) {
    const bool on = c < 4 || c >= 13 || per_rank[0].sel[c - 4] != nullptr;
    if (on) present[np++] = c;
  }
  for (size_t g = 0; g < G; ++g) {
    const plk_circuit_io& c = per_rank[g];
    for (int k = 0; k < 4; ++k) if (!c.wires[k] || !c.sigma[k]) return node_fail(node, PLK_ERR_ARG, "null column pointer");
    for (int k = 0; k < 9; ++k) if ((c.sel[k] != nullptr) != (per_rank[0].sel[k] != nullptr)) return node_fail(node, PLK_ERR_ARG, "the ranks disagree on which selectors are present");
    if (!c.z_out || !c.gate_out || !c.perm_out) return node_fail(node, PLK_ERR_ARG, "null output pointer");
  }
  const size_t slots = (np + G - 1) / G;                    // columns per rank, at most
  const size_t piece = chunk + z_shift;                     // a range's points followed by the next z_shift (halo)
  const size_t block = slots * piece;                       // elements per (source, destination) pair
  int rc = ensure_buffers(node, 5, 6, G * block * ELEM);
  if (rc) return rc;
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    const plk_circuit_io& c = per_rank[g];
    me.rc = PLK_OK; me.err.clear();
    bool ok = HIP_OK(hipSetDevice(me.device));
    hipStream_t s = me.stream;
    char *send = (char*)me.buf[5], *recv = (char*)me.buf[6];
    if (ok && node->have_done)
      for (size_t h = 0; h < G && ok; ++h) ok = HIP_OK(hipStreamWaitEvent(s, node->r[h].done, 0));
    // Z on every rank (a third of a millisecond at 2^20 rows; replicated, it needs no exchange of its own)
    if (ok) ok = PLK_OKAY(plk_dev_grand_product(me.ctx, s, c.wires, c.sigma, n, c.beta, c.gamma, c.z_out));
    // my columns: interpolation and extension, written as [destination][slot][its points + halo]
    const uint64_t* cols[18]; uint32_t kinds[18]; uint64_t* outs[18]; size_t mine = 0;
    for (size_t i = g; i < np; i += G) {
      const int col = present[i];
      cols[mine] = col < 4 ? c.wires[col] : col < 13 ? c.sel[col - 4] : col < 17 ? c.sigma[col - 13] : c.z_out;
      kinds[mine] = (col >= 13 && col < 17) ? PLK_INPUT_U64 : PLK_INPUT_FIELD;
      outs[mine] = (uint64_t*)(send + mine * piece * ELEM);
      ++mine;
    }
    if (ok && mine) ok = PLK_OKAY(plk_dev_extend4(me.ctx, s, cols, kinds, mine, n, nullptr, outs, chunk, block, z_shift));
    // exchange: my slots, every rank's points -> every rank's slots, my points
    if (node->exchange == PLK_NODE_RCCL) {
      if (ok) ok = NCCL_OK(g_rccl.AllToAll(send, recv, block * 4, ncclUint64, me.comm, s));
    } else {
      for (size_t h = 0; h < G && ok; ++h)
        ok = HIP_OK(hipMemcpyAsync((char*)node->r[h].buf[6] + g * block * ELEM, send + h * block * ELEM, block * ELEM, hipMemcpyDefault, s));
      if (ok) ok = HIP_OK(hipEventRecord(me.ev[0], s));
      node->barrier.wait();
      for (size_t h = 0; h < G && ok; ++h)
        if (h != g) ok = HIP_OK(hipStreamWaitEvent(s, node->r[h].ev[0], 0));
    }
    // column `present[i]` came from rank i % G in its slot i / G
    const uint64_t* ev[18] = {nullptr};
    for (size_t i = 0; i < np; ++i) ev[present[i]] = (const uint64_t*)(recv + ((i % G) * block + (i / G) * piece) * ELEM);
    if (ok) ok = PLK_OKAY(plk_dev_gate_constraint_eval(me.ctx, s, ev, ev + 4, chunk, c.gate_out));
    if (ok) ok = PLK_OKAY(plk_dev_perm_constraint_eval_range(me.ctx, s, ev[17], ev + 13, ev, chunk, g * chunk, c.beta, c.gamma, z_shift, c.perm_out));
    if (ok) ok = HIP_OK(hipEventRecord(me.done, s));
    // the data-dependent status (zero denominator, sigma index): waits for this rank's stream
    const int st = plk_dev_status(me.ctx, s);
    if (ok) (void)plk_ok(me, st, "plk_dev_status");
  });
  node->have_done = true;
  return collect(node, "plk_node_round23");
}

// plk_node_round23 on host arrays: the columns go up to every rank's GPU (each over its own link), the pieces of the
// two constraint vectors come down from the rank that computed them, Z from rank 0.
int plk_node_round23_host(plk_node* node, const uint64_t* const wires[4], const uint64_t* const sel[9], const uint64_t* const sigma[4],
                          size_t n, size_t N, const uint64_t beta[4], const uint64_t gamma[4], size_t z_shift, uint64_t* z_out,
                          uint64_t* gate_out, uint64_t* perm_out) {
  if (!node) return PLK_ERR_ARG;
  node->last_error.clear();
  if (!wires || !sigma || !beta || !gamma || !z_out || !gate_out || !perm_out) return node_fail(node, PLK_ERR_ARG, "null pointer");
  for (int k = 0; k < 4; ++k) if (!wires[k] || !sigma[k]) return node_fail(node, PLK_ERR_ARG, "null column pointer");
  const size_t G = node->r.size();
  if (log2_exact(n) < 2 || N != 4 * n || N % G) return node_fail(node, PLK_ERR_ARG, "plk_node_round23_host takes N = 4 n, n a power of two >= 4, and ranks that divide N");
  const size_t piece = N / G, col = n * ELEM, idx = n * sizeof(uint64_t);
  // the slab of one rank: 4 wires, 9 selectors, 4 sigma (u64), Z (n + 1), the two pieces; every part at a multiple of 256 bytes
  auto up = [](size_t b) { return (b + 255) & ~size_t(255); };
  const size_t o_w = 0, o_s = o_w + 4 * up(col), o_g = o_s + 9 * up(col), o_z = o_g + 4 * up(idx), o_gate = o_z + up(col + ELEM),
               o_perm = o_gate + up(piece * ELEM), total = o_perm + up(piece * ELEM);
  int rc = ensure_buffers(node, 7, 7, total);
  if (rc) return rc;
  std::vector<plk_circuit_io> io(G);
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    me.rc = PLK_OK; me.err.clear();
    if (!HIP_OK(hipSetDevice(me.device))) return;
    char* base = (char*)me.buf[7];
    plk_circuit_io& c = io[g];
    bool ok = true;
    for (int k = 0; k < 4 && ok; ++k) {
      c.wires[k] = (const uint64_t*)(base + o_w + k * up(col));
      c.sigma[k] = (const uint64_t*)(base + o_g + k * up(idx));
      ok = PLK_OKAY(plk_dev_upload(me.ctx, me.stream, (void*)c.wires[k], wires[k], col)) &&
           PLK_OKAY(plk_dev_upload(me.ctx, me.stream, (void*)c.sigma[k], sigma[k], idx));
    }
    for (int k = 0; k < 9 && ok; ++k) {
      c.sel[k] = nullptr;
      if (!sel || !sel[k]) continue;
      c.sel[k] = (const uint64_t*)(base + o_s + k * up(col));
      ok = PLK_OKAY(plk_dev_upload(me.ctx, me.stream, (void*)c.sel[k], sel[k], col));
    }
    for (int k = 0; k < 4; ++k) { c.beta[k] = beta[k]; c.gamma[k] = gamma[k]; }
    c.z_out = (uint64_t*)(base + o_z); c.gate_out = (uint64_t*)(base + o_gate); c.perm_out = (uint64_t*)(base + o_perm);
  });
  if ((rc = collect(node, "plk_node_round23_host (upload)"))) return rc;
  if ((rc = plk_node_round23(node, io.data(), n, N, z_shift))) return rc;
  node->workers.run([&](size_t g) {
    Rank& me = node->r[g];
    me.rc = PLK_OK; me.err.clear();
    if (!HIP_OK(hipSetDevice(me.device))) return;
    bool ok = PLK_OKAY(plk_dev_download(me.ctx, me.stream, (char*)gate_out + g * piece * ELEM, io[g].gate_out, piece * ELEM)) &&
              PLK_OKAY(plk_dev_download(me.ctx, me.stream, (char*)perm_out + g * piece * ELEM, io[g].perm_out, piece * ELEM));
    if (ok && g == 0) (void)PLK_OKAY(plk_dev_download(me.ctx, me.stream, z_out, io[0].z_out, (n + 1) * ELEM));
  });
  return collect(node, "plk_node_round23_host (download)");
}

}  // extern "C"
