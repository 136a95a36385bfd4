// csrc/multi_gpu.hip -- the verification loop sharded over the GPUs of one node, behind the C ABI.
//
// north_star / SURVEY 8(e): hypotheses are independent, so the batch is block-partitioned over the
// devices (clouds + index replicated on each), every device scores its contiguous slice into a
// zero-initialised full-length vector, ONE RCCL all-reduce(integer sum over {scores | counts}) over xGMI leaves
// every device with all scores and counts, and the arg-max -- with the exact near-tie settlement of lcp_score.hip --
// is taken on device 0.  The consumers are the per-object loops of the node
// (PPE/data_layer/SceneCfg.cpp:376-406 -> ObjectPoseCandidateSet.cpp:66-68) and the score readers of
// the search (PPE/hypothesis_verification/HypothesisSelection.cpp:248-257): they get the same
// arrays a single-device pgp_score_lcp returns.
//
// One process, one host thread + one stream per device (the worker owns hipSetDevice for its
// thread); the calling thread only copies the transforms into a pinned buffer, posts one job per
// worker, issues the grouped collective and waits for device 0.  RCCL is bound at run time
// (dlopen of librccl.so.1, the library torch's "nccl" backend is): libpgp.so has no link-time
// dependency on it and a single-device group needs no collective at all (PGP_MULTI_FORCE_COLLECTIVE=1
// runs a one-rank communicator anyway -- the GPU test of the exchange path on a 1-GPU box).
//
// PGP_MULTI_EMULATE=n (n >= 2): n logical members on ONE device, each with its own context, worker
// thread and stream -- every N > 1 branch below (zeroed full-length vectors, slice offsets, the
// exchange, settlement and exact records across slices) runs on a 1-GPU box.  RCCL refuses a
// communicator with a device listed twice, so the exchange is a sum kernel of this file with the
// all-reduce's semantics (every member ends up with the element-wise sum); everything else is the
// production code path.  A test vehicle, never a performance configuration.

#include "pgp_internal.h"
#include "host_worker.h"

#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace pgp {
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl* r) {
  if (r->handle) return true;
  static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    r->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (r->handle) break;
  }
  if (!r->handle) {
    set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror());
    return false;
  }
#define PGP_SYM(field, name)                                             \
  r->field = reinterpret_cast<decltype(r->field)>(dlsym(r->handle, name)); \
  if (!r->field) {                                                       \
    set_error("RCCL symbol %s missing", name);                           \
    return false;                                                        \
  }
  PGP_SYM(CommInitAll, "ncclCommInitAll")
  PGP_SYM(CommInitRank, "ncclCommInitRank")
  PGP_SYM(GetUniqueId, "ncclGetUniqueId")
  PGP_SYM(CommCount, "ncclCommCount")
  PGP_SYM(CommDestroy, "ncclCommDestroy")
  PGP_SYM(AllReduce, "ncclAllReduce")
  PGP_SYM(GroupStart, "ncclGroupStart")
  PGP_SYM(GroupEnd, "ncclGroupEnd")
  PGP_SYM(GetErrorString, "ncclGetErrorString")
#undef PGP_SYM
  return true;
}

// (Worker: one host thread per device, csrc/host_worker.h -- HIP-free so that it also builds under ThreadSanitizer)

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace
}  // namespace pgp

using namespace pgp;

// The streaming form's per-member state (pgp_multi_enqueue_slot): a second stream for the exchange and two {scores | counts}
// vectors, so that the all-reduce of step i runs under the scoring of step i + 1.
struct Streaming {
  hipStream_t x = nullptr;
  DevBuf ring[2];
  hipEvent_t scored[2] = {nullptr, nullptr}, reduced[2] = {nullptr, nullptr};
};
// PGP_MULTI_EMULATE_RANKED=1: the exchange of a group that spans PROCESSES without RCCL (which refuses one device twice):
// every rank's slice goes through a POSIX shared-memory segment named after the group's id -- so that the launcher form (one
// process per rank, slices by global rank, every process holding all transforms and taking the arg-max itself) runs with
// several ranks on ONE device.  Host-synchronous, a test vehicle like PGP_MULTI_EMULATE, never a performance configuration.
struct ShmExchange {
  struct Header {
    std::atomic<unsigned> arrived;
    unsigned pad[15];
  };
  Header* h = nullptr;
  unsigned char* data = nullptr;   // two buffers of `cap` bytes
  size_t cap = 0, total = 0;
  std::string name;
  unsigned seq = 0;
};
constexpr size_t kShmCap = (size_t)16 << 20;   // per buffer: 2 M hypotheses

constexpr int kSlots = 16;        // resident hypothesis batches of the streaming form
constexpr int kEmulateMax = 16;   // members of an emulated group the streaming form's sum kernel takes

struct pgp_multi {
  int n = 0;                      // members in THIS process
  int world = 0, rank0 = 0;       // member k is rank rank0 + k of `world` (single-process group: world = n, rank0 = 0)
  long long n_exchanges = 0;      // exchanges (all-reduces) issued per member since the group was created
  std::vector<int> dev;
  // [object][member]: an object = one (scene, model) pair -- a segment of the frame and the object model it is
  // matched against (SceneCfg.cpp:376-406 loops over them) -- replicated on every member.  Object 0 exists from
  // pgp_multi_create on (the single-object entry points act on it); pgp_multi_add_object appends.
  std::vector<std::vector<pgp_ctx*>> octx;
  std::vector<hipStream_t> stream;
  std::vector<Worker*> worker;
  bool use_coll = false;
  bool grouped = false;   // PGP_MULTI_COLL=grouped: the collective is issued for all members by the calling thread
  bool emulate = false;   // PGP_MULTI_EMULATE: members share one device, exchange = emulate_sum
  bool emulate_ranked = false;   // PGP_MULTI_EMULATE_RANKED: a ranked group's exchange through shared memory (ShmExchange)
  ShmExchange shm;
  std::vector<hipEvent_t> ev;   // emulate: one event per member
  DevBuf d_sum;                 // emulate: the summed vector before it is handed to every member
  Rccl rccl;
  std::vector<ncclComm_t> comm;
  // per device: the transform list (member 0: all of it; the others: their slice, in place) and the full-length
  // {scores | counts} vector over the FLAT (object, hypothesis) space
  std::vector<DevBuf> d_T, d_all, d_best;
  std::vector<int> off{0, 0};   // uploaded hypotheses: object o owns flat positions off[o] .. off[o + 1] - 1
  void* h_pin = nullptr;  // portable pinned staging: transforms in, scores | counts | best out
  size_t h_pin_cap = 0;
  float last_ms[3] = {0.f, 0.f, 0.f};  // host wall clock of the last call: upload, enqueue, total
  // ICP pose shards: per member one context per job slot (a context keeps ONE target index, icp.hip)
  std::vector<std::vector<pgp_ctx*>> ictx;
  // congruent sets sharded by base: per object the base boundaries of the last pgp_multi_find_congruent_batch
  std::vector<std::vector<int>> cs_lo;
  // emulate: the members' worker threads meet here (the exchange kernel needs every slice queued)
  std::mutex bar_mu;
  std::condition_variable bar_cv;
  int bar_count = 0, bar_gen = 0;

  // streaming form
  std::vector<Streaming> s2;                 // [member]
  std::vector<std::vector<DevBuf>> slot_T;   // [slot][member]: the transforms of a resident batch (member 0: all of it)
  std::vector<int> slot_n;                   // [slot] hypotheses, -1 = empty
  DevBuf d_best2;                            // member 0: {index, score bits} of ring[0] and ring[1]
  DevBuf d_tail_seq;                         // member 0: the settlement's workspace of the tails (they run on the exchange stream)
  // member 0 scores its slice like everybody else but needs ALL transforms for the arg-max's near-tie settlement: the ones
  // outside its slice go up on a side stream, beside its scoring; the tail waits for ev_rest
  hipStream_t up_stream = nullptr;
  hipEvent_t ev_rest = nullptr;
  bool rest_pending = false;
  hipEvent_t ev_sum[2] = {nullptr, nullptr}; // emulate: the summed vector of ring[b] has been handed to every member
  long long step = 0;                        // steps enqueued since the last collect
  int pend_slot[2] = {0, 0}, pend_mode[2] = {0, 0}, pend_n[2] = {0, 0};
  float pend_gate[2] = {0.f, 0.f};
  bool exchange() const { return world > 1 || use_coll; }

  int total() const { return off.back(); }
  int objects() const { return (int)octx.size(); }
  void barrier() {
    std::unique_lock<std::mutex> lk(bar_mu);
    const int gen = bar_gen;
    if (++bar_count == n) {
      bar_count = 0;
      ++bar_gen;
      bar_cv.notify_all();
    } else {
      bar_cv.wait(lk, [&] { return bar_gen != gen; });
    }
  }
};

namespace {

// out[i] = sum over members of in[k][i] over the 2 n_h 32-bit words {scores | counts}, as INTEGERS -- what the one
// RCCL all-reduce of the production path computes (every element is non-zero in at most one member, so the sum of
// the bit patterns is that member's pattern, for the float scores too)
__global__ __launch_bounds__(256) void emulate_sum(const float* const* __restrict__ in, int n_members, int n_h,
                                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n_h) return;
  int s = 0;
  for (int k = 0; k < n_members; ++k) s += reinterpret_cast<const int*>(in[k])[i];
  reinterpret_cast<int*>(out)[i] = s;
}

// member 0's job with its device current on the CALLING thread (restored afterwards)
int on_caller(pgp_multi* m, const std::function<int()>& job) {
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  if (prev != m->dev[0] && hipSetDevice(m->dev[0]) != hipSuccess) {
    set_error("hipSetDevice(%d) failed", m->dev[0]);
    return PGP_EHIP;
  }
  const int r = job();
  if (prev >= 0 && prev != m->dev[0]) (void)hipSetDevice(prev);
  return r;
}

// fn(k) for every member: members 1 .. n-1 on their worker threads, member 0 -- the one whose work ends the call: the arg-max
// and the way home -- on the calling thread itself: a hand-off to a worker and back costs the call ~20 us (0.146 against
// 0.127 ms for a one-member group's host-pointer call), and a group of one then hands nothing off at all.
// PGP_MULTI_CALLER_RUNS=0: every member on its worker (the form up to round 5).
int run_all(pgp_multi* m, const std::function<int(int)>& fn) {
  static const bool caller_runs = !(getenv("PGP_MULTI_CALLER_RUNS") && atoi(getenv("PGP_MULTI_CALLER_RUNS")) == 0);
  const int first = caller_runs ? 1 : 0;
  for (int k = first; k < m->n; ++k) m->worker[k]->post([&fn, k] { return fn(k); });
  int rc = PGP_OK;
  if (caller_runs) {
    char keep[512] = "";
    rc = on_caller(m, [&fn] { return fn(0); });
    if (rc != PGP_OK) {   // (the text is this thread's own already: keep it across the waits below)
      std::strncpy(keep, pgp_last_error(), sizeof keep - 1);
      keep[sizeof keep - 1] = 0;
    }
    for (int k = 1; k < m->n; ++k) {
      const int r = m->worker[k]->wait();
      if (r != PGP_OK && rc == PGP_OK) {
        rc = r;
        set_error("device %d: %s", m->dev[k], m->worker[k]->err);
      }
    }
    if (keep[0]) set_error("device %d: %s", m->dev[0], keep);
    return rc;
  }
  for (int k = 0; k < m->n; ++k) {
    int r = m->worker[k]->wait();
    if (r != PGP_OK && rc == PGP_OK) {
      rc = r;
      set_error("device %d: %s", m->dev[k], m->worker[k]->err);
    }
  }
  return rc;
}

int ensure_pin(pgp_multi* m, size_t bytes) {
  if (bytes <= m->h_pin_cap) return PGP_OK;
  if (m->h_pin) {
    (void)hipHostFree(m->h_pin);
    m->h_pin = nullptr;
    m->h_pin_cap = 0;
  }
  const size_t want = bytes + bytes / 4 + 256;
  PGP_HIP(hipHostMalloc(&m->h_pin, want, hipHostMallocPortable));
  m->h_pin_cap = want;
  return PGP_OK;
}

bool bad_object(const pgp_multi* m, int obj, const char* who) {
  if (!m) {
    set_error("%s: handle is NULL", who);
    return true;
  }
  if (obj < 0 || obj >= m->objects()) {
    set_error("%s: object %d of %d", who, obj, m->objects());
    return true;
  }
  return false;
}

void slice_of(int n_total, int k, int n_dev, int* lo, int* hi) {
  // contiguous slices, sizes differ by at most one, earlier devices larger (sharding.shard_bounds)
  const int base = n_total / n_dev, rem = n_total % n_dev;
  *lo = k * base + (k < rem ? k : rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
}

// The all-reduce of a ranked group under PGP_MULTI_EMULATE_RANKED: this rank's slice of {scores | counts} into the shared
// segment, a barrier over the ranks, the complete vector back (= the sum: every element is non-zero on its owner only).
int shm_exchange(pgp_multi* m, float* d_s, int N, hipStream_t st) {
  ShmExchange& x = m->shm;
  const size_t bytes = (size_t)N * 8;
  if (bytes > x.cap) {
    set_error("PGP_MULTI_EMULATE_RANKED: %d hypotheses exceed the shared segment", N);
    return PGP_EINVAL;
  }
  int lo, hi;
  slice_of(N, m->rank0, m->world, &lo, &hi);
  unsigned char* buf = x.data + (size_t)(x.seq & 1u) * x.cap;
  if (hi > lo) {
    PGP_HIP(hipMemcpyAsync(buf + (size_t)lo * 4, d_s + lo, (size_t)(hi - lo) * 4, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipMemcpyAsync(buf + (size_t)N * 4 + (size_t)lo * 4, reinterpret_cast<int*>(d_s + N) + lo, (size_t)(hi - lo) * 4,
                           hipMemcpyDeviceToHost, st));
  }
  PGP_HIP(hipStreamSynchronize(st));
  x.h->arrived.fetch_add(1u, std::memory_order_acq_rel);
  const unsigned want = (unsigned)m->world * (x.seq + 1u);
  const auto t_end = std::chrono::steady_clock::now() + std::chrono::seconds(120);
  while (x.h->arrived.load(std::memory_order_acquire) < want) {
    if (std::chrono::steady_clock::now() >= t_end) {
      set_error("PGP_MULTI_EMULATE_RANKED: a rank did not arrive at exchange %u", x.seq);
      return PGP_EHIP;
    }
    std::this_thread::yield();
  }
  PGP_HIP(hipMemcpyAsync(d_s, buf, bytes, hipMemcpyHostToDevice, st));
  PGP_HIP(hipStreamSynchronize(st));   // (the buffer is written again two exchanges on, behind the next one's barrier)
  ++x.seq;
  return PGP_OK;
}

// The scoring call over whatever pgp_multi_upload[_objects] left on the devices.  scores / counts: flat, object after
// object; best_index / best_score: one entry per object.
int score_flat(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts, int* best_index, float* best_score) {
  const int N = m->total(), n_obj = (int)m->off.size() - 1;
  const double t0 = now_ms();
  int rc = ensure_pin(m, (size_t)N * 72 + (size_t)n_obj * 8 + 256);   // nothing uploaded yet: an empty batch still returns {-1, 0}
  if (rc != PGP_OK) return rc;
  unsigned char* pin_out = static_cast<unsigned char*>(m->h_pin) + (((size_t)N * 64 + 63) & ~(size_t)63);
  std::vector<int> fail((size_t)m->n, 0);   // emulate: a member that failed to queue its slice (the others still meet it)
  double t_enq = t0;

  // every member: its pieces of the flat space into a zeroed full-length vector (the sum over members is the gather)
  auto score_slice = [&](int k) -> int {
    int lo, hi;
    slice_of(N, m->rank0 + k, m->world, &lo, &hi);
    float* d_s = m->d_all[k].as<float>();
    int* d_c = reinterpret_cast<int*>(d_s + N);
    if (N > 0 && m->world > 1) PGP_HIP(hipMemsetAsync(d_s, 0, (size_t)N * 8, m->stream[k]));
    for (int o = 0; o < n_obj; ++o) {
      const int a = std::max(lo, m->off[o]), b = std::min(hi, m->off[o + 1]);
      if (b <= a) continue;
      pgp_ctx* c = m->octx[o][k];
      // the exact-records pass (pgp_set_exact_records on member 0's context) belongs to the COMPLETE vector, below;
      // Verify's early termination depends on ALL earlier hypotheses of the object: below as well
      const bool records = c->exact_records, early = c->verify_early_out;
      c->exact_records = false;
      c->verify_early_out = false;
      const int r = pgp_score_lcp_device(c, m->d_T[k].as<float>() + 16 * (size_t)a, b - a, mode, gate_deg, d_s + a, d_c + a,
                                         nullptr, m->stream[k]);
      c->exact_records = records;
      c->verify_early_out = early;
      if (r != PGP_OK) return r;
    }
    return PGP_OK;
  };
  // member 0: per object the arg-max over its complete vector (exact under weighted near-ties), one copy back
  auto tail = [&]() -> int {
    float* d_s = m->d_all[0].as<float>();
    int* d_b = m->d_best[0].as<int>();
    hipStream_t st = m->stream[0];
    if (m->rest_pending) PGP_HIP(hipStreamWaitEvent(st, m->ev_rest, 0));   // the transforms outside member 0's slice (upload_flat)
    for (int o = 0; o < n_obj; ++o) {
      pgp_ctx* c = m->octx[o][0];
      const int a = m->off[o], cnt = m->off[o + 1] - a;
      const float* d_To = m->d_T[0].as<float>() + 16 * (size_t)a;
      int r = pgp_settle_best_device(c, d_To, cnt, mode, gate_deg, d_s + a, d_b + 2 * o, st);
      if (r != PGP_OK) return r;
      // pgp_set_exact_records / pgp_set_verify_early_out on member 0's context of the object cover the group's calls
      if (c->exact_records && (r = pgp_settle_records_device(c, d_To, cnt, mode, gate_deg, d_s + a, st)) != PGP_OK) return r;
      if (c->verify_early_out && mode == PGP_MODE_PLAIN && cnt > 0 &&
          (r = pgp_verify_early_out_device(c, d_To, cnt, d_s + a, reinterpret_cast<int*>(d_s + N) + a, st)) != PGP_OK)
        return r;
    }
    t_enq = now_ms();
    // home: by ONE kernel and a completion word while the arrays are small (pgp::publish_and_wait), else two copies and the stream
    const PubItem items[2] = {{d_s, pin_out, (size_t)N * 8}, {d_b, pin_out + (size_t)N * 8, (size_t)n_obj * 8}};
    const PubItem* first = N > 0 ? items : items + 1;
    const int n_items = N > 0 ? 2 : 1;
    if (publish_usable(first, n_items)) return publish_and_wait(m->octx[0][0], st, first, n_items);
    if (N > 0) PGP_HIP(hipMemcpyAsync(pin_out, d_s, (size_t)N * 8, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipMemcpyAsync(pin_out + (size_t)N * 8, d_b, (size_t)n_obj * 8, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipStreamSynchronize(st));
    return PGP_OK;
  };
  // the exchange on ONE device (PGP_MULTI_EMULATE): member 0's stream waits for every slice, sums the vectors and hands
  // the sum to every member, whose streams then wait for it
  auto emulate_exchange = [&](int k, int rc_mine) -> int {
    if (rc_mine == PGP_OK && hipEventRecord(m->ev[k], m->stream[k]) != hipSuccess) rc_mine = PGP_EHIP;
    fail[(size_t)k] = rc_mine != PGP_OK;
    m->barrier();
    bool any = false;
    for (int f : fail) any = any || f;
    int r = PGP_OK;
    if (k == 0 && !any) {
      r = [&]() -> int {
        hipStream_t st = m->stream[0];
        int q;
        if ((q = m->d_sum.ensure((size_t)N * 8 + (size_t)m->n * sizeof(float*) + 64)) != PGP_OK) return q;
        float* d_out = m->d_sum.as<float>();
        const float** d_ptrs = reinterpret_cast<const float**>(m->d_sum.as<unsigned char>() + (((size_t)N * 8 + 15) & ~(size_t)15));
        std::vector<const float*> ptrs((size_t)m->n);
        for (int j = 0; j < m->n; ++j) {
          ptrs[(size_t)j] = m->d_all[j].as<float>();
          if (j > 0) PGP_HIP(hipStreamWaitEvent(st, m->ev[j], 0));
        }
        PGP_HIP(hipMemcpyAsync(d_ptrs, ptrs.data(), (size_t)m->n * sizeof(float*), hipMemcpyHostToDevice, st));
        PGP_HIP(hipStreamSynchronize(st));   // ptrs is a stack temporary
        hipLaunchKernelGGL(emulate_sum, dim3((2 * N + 255) / 256), dim3(256), 0, st, d_ptrs, m->n, N, d_out);
        PGP_HIP(hipGetLastError());
        for (int j = 0; j < m->n; ++j)
          PGP_HIP(hipMemcpyAsync(m->d_all[j].p, d_out, (size_t)N * 8, hipMemcpyDeviceToDevice, st));
        PGP_HIP(hipEventRecord(m->ev[0], st));
        return PGP_OK;
      }();
      fail[0] = r != PGP_OK;
    }
    m->barrier();
    if (rc_mine != PGP_OK) return rc_mine;
    if (r != PGP_OK) return r;
    if (any || fail[0]) {
      set_error("another member of the group failed");
      return PGP_EHIP;
    }
    if (k > 0) PGP_HIP(hipStreamWaitEvent(m->stream[k], m->ev[0], 0));
    return PGP_OK;
  };
  // ONE all-reduce per call over the 8 N bytes {scores | counts}, summed as 32-bit integers: every element is non-zero
  // on exactly one member (its owner) and all-zero bits elsewhere, and x + 0 + ... + 0 over the BIT PATTERNS returns
  // x's pattern -- exact for the float scores too (scores are >= +0: no -0, no NaN).  At 4096 hypotheses the exchange is
  // latency-bound (32 KB per member), so two collectives cost twice what one does.
  auto own_allreduce = [&](int k) -> int {
    float* d_s = m->d_all[k].as<float>();
    if (m->emulate_ranked) return shm_exchange(m, d_s, N, m->stream[k]);
    const ncclResult_t nr = m->rccl.AllReduce(d_s, d_s, 2 * (size_t)N, ncclInt32, ncclSum, m->comm[k], m->stream[k]);
    if (nr != ncclSuccess) {
      set_error("ncclAllReduce failed: %s", m->rccl.GetErrorString(nr));
      return PGP_EHIP;
    }
    return PGP_OK;
  };

  const bool exchange = m->exchange();
  if (exchange && N > 0) ++m->n_exchanges;
  if (m->emulate && m->n > 1 && N > 0) {
    rc = run_all(m, [&](int k) -> int {
      int r = emulate_exchange(k, score_slice(k));
      if (r == PGP_OK && k == 0) r = tail();
      return r;
    });
  } else if (exchange && m->use_coll && !m->emulate && N > 0 && (!m->grouped || m->emulate_ranked)) {
    // default: every member's worker thread queues its slice, its own all-reduce call (one communicator per device, the
    // multi-thread form RCCL documents) and, on member 0, the tail -- ONE rendezvous of the calling thread per call.
    // A member whose slice failed to queue still joins the collective: the others' streams would wait for it forever.
    rc = run_all(m, [&](int k) -> int {
      const int r1 = score_slice(k);
      const int r2 = own_allreduce(k);
      if (r1 != PGP_OK) return r1;
      if (r2 != PGP_OK) return r2;
      return k == 0 ? tail() : PGP_OK;
    });
  } else {
    // PGP_MULTI_COLL=grouped (and the single-member group without a collective): slices, then the collective for all
    // members from the calling thread inside one group, then the tail
    rc = run_all(m, score_slice);
    if (rc != PGP_OK) return rc;
    if (m->use_coll && !m->emulate && N > 0) {
      ncclResult_t nr = m->rccl.GroupStart();
      for (int k = 0; k < m->n && nr == ncclSuccess; ++k) {
        float* d_s = m->d_all[k].as<float>();
        nr = m->rccl.AllReduce(d_s, d_s, 2 * (size_t)N, ncclInt32, ncclSum, m->comm[k], m->stream[k]);
      }
      ncclResult_t ge = m->rccl.GroupEnd();
      if (nr == ncclSuccess) nr = ge;
      if (nr != ncclSuccess) {
        set_error("ncclAllReduce failed: %s", m->rccl.GetErrorString(nr));
        return PGP_EHIP;
      }
    }
    rc = on_caller(m, tail);
  }
  if (rc != PGP_OK) return rc;
  if (N > 0) {
    if (scores) std::memcpy(scores, pin_out, (size_t)N * 4);
    if (counts) std::memcpy(counts, pin_out + (size_t)N * 4, (size_t)N * 4);
  }
  for (int o = 0; o < n_obj; ++o) {
    int best[2];
    std::memcpy(best, pin_out + (size_t)N * 8 + (size_t)o * 8, sizeof best);
    if (best_index) best_index[o] = best[0];
    if (best_score) std::memcpy(best_score + o, &best[1], 4);
  }
  m->last_ms[1] = (float)(t_enq - t0);
  m->last_ms[2] = (float)(now_ms() - t0);
  return PGP_OK;
}

// transforms of n_obj objects -> pinned image -> devices; sets m->off
int upload_flat(pgp_multi* m, const float* const* T, const int* n_h, int n_obj) {
  std::vector<int> off((size_t)n_obj + 1, 0);
  for (int o = 0; o < n_obj; ++o) {
    if (n_h[o] < 0 || (n_h[o] > 0 && !T[o])) {
      set_error("pgp_multi upload: bad hypothesis list of object %d", o);
      return PGP_EINVAL;
    }
    if ((long long)off[o] + n_h[o] > 0x3FFFFFFF) {
      set_error("pgp_multi upload: more than 2^30 hypotheses");
      return PGP_EINVAL;
    }
    off[o + 1] = off[o] + n_h[o];
  }
  const int N = off[n_obj];
  const size_t nT = (size_t)N * 64;
  if (m->rest_pending) {   // (the previous list's side-stream copies read the pinned image that is rewritten below)
    PGP_HIP(hipEventSynchronize(m->ev_rest));
    m->rest_pending = false;
  }
  int rc = ensure_pin(m, nT + (size_t)N * 8 + (size_t)n_obj * 8 + 256);
  if (rc != PGP_OK) return rc;
  for (int o = 0; o < n_obj; ++o)
    if (n_h[o]) std::memcpy(static_cast<unsigned char*>(m->h_pin) + (size_t)off[o] * 64, T[o], (size_t)n_h[o] * 64);
  m->off = off;
  rc = run_all(m, [m, nT, N, n_obj](int k) -> int {
    int lo, hi, r;
    slice_of(N, m->rank0 + k, m->world, &lo, &hi);
    if ((r = m->d_T[k].ensure(nT)) != PGP_OK) return r;
    if ((r = m->d_all[k].ensure((size_t)N * 8)) != PGP_OK) return r;
    if ((r = m->d_best[k].ensure((size_t)n_obj * 8 + 16)) != PGP_OK) return r;
    for (int o = 0; o < n_obj; ++o) {
      // member 0 also works on every object's complete vector (settlement, exact records, Verify's early termination)
      const int a = std::max(lo, m->off[o]), b = std::min(hi, m->off[o + 1]);
      const int need = k == 0 ? m->off[o + 1] - m->off[o] : std::max(b - a, 0);
      if ((r = pgp_reserve(m->octx[o][k], need)) != PGP_OK) return r;
    }
    // member 0 holds ALL transforms (it settles near-ties across slices); the others copy their slice only, to the
    // place it has in the flat list (64 B per hypothesis: 4 MB at 65 536 -- every device over its own PCIe link)
    const size_t a = (size_t)lo * 64, b = (size_t)hi * 64;
    unsigned char* dst = m->d_T[k].as<unsigned char>();
    unsigned char* src = static_cast<unsigned char*>(m->h_pin);
    // (out of the portable pinned image by a kernel on the member's stream: no copy-engine hand-over in front of the scoring)
    if (b > a && (r = stage_to_device(m->stream[k], dst + a, src + a, b - a)) != PGP_OK) return r;
    if (k == 0 && (a > 0 || b < nT)) {
      // the rest of the list, for the settlement only: on the side stream, under this member's scoring (at eight members
      // 7/8 of 2 MB: ~35 us that were in front of the scoring launch)
      if (!m->up_stream) {
        PGP_HIP(hipStreamCreateWithFlags(&m->up_stream, hipStreamNonBlocking));
        PGP_HIP(hipEventCreateWithFlags(&m->ev_rest, hipEventDisableTiming));
      }
      if (a > 0 && (r = stage_to_device(m->up_stream, dst, src, a)) != PGP_OK) return r;
      if (b < nT && (r = stage_to_device(m->up_stream, dst + b, src + b, nT - b)) != PGP_OK) return r;
      PGP_HIP(hipEventRecord(m->ev_rest, m->up_stream));
      m->rest_pending = true;
    }
    return PGP_OK;
  });
  if (rc != PGP_OK) m->off.assign(2, 0);   // nothing usable is resident: a later *_uploaded call scores the empty batch
  return rc;
}

}  // namespace

extern "C" {

int pgp_multi_slice(int n_total, int k, int n_dev, int* lo, int* hi) {
  if (n_total < 0 || n_dev <= 0 || k < 0 || k >= n_dev || !lo || !hi) {
    set_error("pgp_multi_slice: bad argument");
    return PGP_EINVAL;
  }
  slice_of(n_total, k, n_dev, lo, hi);
  return PGP_OK;
}

int pgp_multi_flat_slices(const int* n_h, int n_obj, int k, int n_dev, int* obj, int* lo, int* hi, int* n_pieces) {
  if (n_obj < 0 || (n_obj > 0 && (!n_h || !obj || !lo || !hi)) || n_dev <= 0 || k < 0 || k >= n_dev || !n_pieces) {
    set_error("pgp_multi_flat_slices: bad argument");
    return PGP_EINVAL;
  }
  long long total = 0;
  for (int o = 0; o < n_obj; ++o) {
    if (n_h[o] < 0) {
      set_error("pgp_multi_flat_slices: negative count");
      return PGP_EINVAL;
    }
    total += n_h[o];
  }
  if (total > 0x3FFFFFFF) {
    set_error("pgp_multi_flat_slices: more than 2^30 units");
    return PGP_EINVAL;
  }
  int a, b, base = 0, np = 0;
  slice_of((int)total, k, n_dev, &a, &b);
  for (int o = 0; o < n_obj; ++o) {
    const int x = std::max(a, base), y = std::min(b, base + n_h[o]);
    if (y > x) {
      obj[np] = o;
      lo[np] = x - base;
      hi[np] = y - base;
      ++np;
    }
    base += n_h[o];
  }
  *n_pieces = np;
  return PGP_OK;
}

namespace {
// rank0 < 0: a single-process group (world = its members); else the members are ranks rank0 .. rank0 + n_dev - 1 of `world`
// and `id` is the 128-byte ncclUniqueId every process of the group was given
int create_group(pgp_multi** out, const int* device_ids, int n_dev, int rank0, int world, const void* id) {
  if (!out) {
    set_error("pgp_multi_create: out is NULL");
    return PGP_EINVAL;
  }
  *out = nullptr;
  int visible = 0;
  hipError_t e = hipGetDeviceCount(&visible);
  if (e != hipSuccess || visible <= 0) {
    set_error("no HIP device available (%s); libpgp has no CPU fallback",
              e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return PGP_ENODEV;
  }
  const bool ranked = rank0 >= 0;
  int emulate = 0;
  if (const char* v = getenv("PGP_MULTI_EMULATE")) emulate = atoi(v);
  if (ranked && emulate >= 2) {
    set_error("pgp_multi_create_ranked: PGP_MULTI_EMULATE applies to single-process groups");
    return PGP_EINVAL;
  }
  if (emulate >= 2) n_dev = emulate;   // n logical members on the first listed device
  if (n_dev <= 0) n_dev = visible;  // every visible device
  if (ranked && (world < n_dev || rank0 + n_dev > world || !id)) {
    set_error("pgp_multi_create_ranked: ranks %d .. %d of %d", rank0, rank0 + n_dev - 1, world);
    return PGP_EINVAL;
  }
  pgp_multi* m = new pgp_multi();
  m->n = n_dev;
  m->world = ranked ? world : n_dev;
  m->rank0 = ranked ? rank0 : 0;
  m->emulate = emulate >= 2;
  for (int k = 0; k < n_dev; ++k) {
    const int d = m->emulate ? (device_ids ? device_ids[0] : 0) : (device_ids ? device_ids[k] : k);
    if (d < 0 || d >= visible) {
      set_error("device %d out of range (%d devices)", d, visible);
      delete m;
      return PGP_EINVAL;
    }
    for (int j = 0; j < k && !m->emulate; ++j)
      if (m->dev[j] == d) {
        set_error("device %d listed twice", d);
        delete m;
        return PGP_EINVAL;
      }
    m->dev.push_back(d);
  }
  m->octx.assign(1, std::vector<pgp_ctx*>((size_t)n_dev, nullptr));
  m->cs_lo.assign(1, std::vector<int>());
  m->ictx.assign((size_t)n_dev, std::vector<pgp_ctx*>());
  m->stream.assign(n_dev, nullptr);
  m->d_T.resize(n_dev);
  m->d_all.resize(n_dev);
  m->d_best.resize(n_dev);
  m->s2.resize((size_t)n_dev);
  m->slot_T.assign(kSlots, std::vector<DevBuf>((size_t)n_dev));
  m->slot_n.assign(kSlots, -1);
  int rc = PGP_OK;
  for (int k = 0; k < n_dev && rc == PGP_OK; ++k) {
    Worker* w = new Worker();
    w->device = m->dev[k];
    w->on_start = [w] { (void)hipSetDevice(w->device); };
    w->last_error = [] { return pgp_last_error(); };
    w->th = std::thread([w] { w->loop(); });
    m->worker.push_back(w);
  }
  rc = run_all(m, [m](int k) -> int {
    int r = pgp_create(&m->octx[0][k], m->dev[k]);
    if (r != PGP_OK) return r;
    PGP_HIP(hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking));
    return m->d_best[k].ensure(16);
  });
  const char* force = getenv("PGP_MULTI_FORCE_COLLECTIVE");
  m->use_coll = m->world > 1 || (force && atoi(force) != 0);
  if (const char* v = getenv("PGP_MULTI_COLL")) m->grouped = std::strcmp(v, "grouped") == 0;
  if (rc == PGP_OK && m->emulate) {
    m->ev.assign(n_dev, nullptr);
    rc = run_all(m, [m](int k) -> int {
      PGP_HIP(hipEventCreateWithFlags(&m->ev[k], hipEventDisableTiming));
      return PGP_OK;
    });
  }
  if (rc == PGP_OK && ranked && getenv("PGP_MULTI_EMULATE_RANKED") && atoi(getenv("PGP_MULTI_EMULATE_RANKED")) != 0) {
    if (n_dev != 1) {
      set_error("PGP_MULTI_EMULATE_RANKED: one member per process");
      rc = PGP_EINVAL;
    } else {
      unsigned long long hsh = 0xCBF29CE484222325ull;
      for (int i = 0; i < 128; ++i) hsh = (hsh ^ static_cast<const unsigned char*>(id)[i]) * 0x100000001B3ull;
      char name[64];
      std::snprintf(name, sizeof name, "/pgp_emu_%016llx", hsh);
      ShmExchange& x = m->shm;
      x.name = name;
      x.cap = kShmCap;
      x.total = 4096 + 2 * x.cap;
      const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
      void* p = MAP_FAILED;
      if (fd >= 0 && ftruncate(fd, (off_t)x.total) == 0) p = mmap(nullptr, x.total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      if (fd >= 0) close(fd);
      if (p == MAP_FAILED) {
        set_error("PGP_MULTI_EMULATE_RANKED: shared segment %s: %s", name, std::strerror(errno));
        rc = PGP_EHIP;
      } else {
        x.h = static_cast<ShmExchange::Header*>(p);   // (a fresh segment is zero-filled: arrived = 0)
        x.data = static_cast<unsigned char*>(p) + 4096;
        m->emulate_ranked = true;
      }
    }
  }
  if (rc == PGP_OK && m->use_coll && !m->emulate && !m->emulate_ranked) {
    if (!load_rccl(&m->rccl)) rc = PGP_ENODEV;
    if (rc == PGP_OK) {
      m->comm.assign(n_dev, nullptr);
      ncclResult_t nr = ncclSuccess;
      if (!ranked) {
        nr = m->rccl.CommInitAll(m->comm.data(), n_dev, m->dev.data());
      } else {
        // one communicator per local member, joined to the ranks of the other processes through the shared id; several
        // members of one process are initialised inside ONE group (the form RCCL documents for one thread, many devices)
        ncclUniqueId uid;
        std::memcpy(&uid, id, sizeof uid);
        int keep = 0;
        (void)hipGetDevice(&keep);
        if (n_dev > 1) nr = m->rccl.GroupStart();
        for (int k = 0; k < n_dev && nr == ncclSuccess; ++k) {
          if (hipSetDevice(m->dev[k]) != hipSuccess) {
            nr = ncclUnhandledCudaError;
            break;
          }
          nr = m->rccl.CommInitRank(&m->comm[(size_t)k], m->world, uid, m->rank0 + k);
        }
        if (n_dev > 1) {
          const ncclResult_t ge = m->rccl.GroupEnd();
          if (nr == ncclSuccess) nr = ge;
        }
        (void)hipSetDevice(keep);
      }
      if (nr != ncclSuccess) {
        set_error("%s failed: %s", ranked ? "ncclCommInitRank" : "ncclCommInitAll", m->rccl.GetErrorString(nr));
        m->comm.clear();
        rc = PGP_EHIP;
      }
    }
  }
  if (rc != PGP_OK) {
    char keep[512];
    std::strncpy(keep, pgp_last_error(), sizeof keep - 1);
    keep[sizeof keep - 1] = 0;
    pgp_multi_destroy(m);
    set_error("%s", keep);
    return rc;
  }
  *out = m;
  return PGP_OK;
}

bool single_process(const pgp_multi* m, const char* who) {
  if (m->world == m->n) return true;
  set_error("%s: the group spans several processes (ranks %d .. %d of %d); this entry point gathers into the caller's "
            "arrays and needs a single-process group", who, m->rank0, m->rank0 + m->n - 1, m->world);
  return false;
}
}  // namespace

int pgp_multi_create(pgp_multi** out, const int* device_ids, int n_dev) {
  return create_group(out, device_ids, n_dev, -1, 0, nullptr);
}

int pgp_multi_unique_id(void* id128) {
  if (!id128) {
    set_error("pgp_multi_unique_id: id128 is NULL");
    return PGP_EINVAL;
  }
  if (getenv("PGP_MULTI_EMULATE_RANKED") && atoi(getenv("PGP_MULTI_EMULATE_RANKED")) != 0) {
    // (no communicator will be made of it: 128 random bytes name the ranks' shared segment)
    std::random_device rd;
    unsigned char* out = static_cast<unsigned char*>(id128);
    for (int i = 0; i < 128; i += 4) {
      const unsigned v = rd();
      std::memcpy(out + i, &v, 4);
    }
    return PGP_OK;
  }
  static Rccl r;   // (the handle stays loaded for the life of the process, as a group's does)
  if (!load_rccl(&r)) return PGP_ENODEV;
  ncclUniqueId uid;
  const ncclResult_t nr = r.GetUniqueId(&uid);
  if (nr != ncclSuccess) {
    set_error("ncclGetUniqueId failed: %s", r.GetErrorString(nr));
    return PGP_EHIP;
  }
  static_assert(sizeof uid == 128, "pgp.h promises 128 bytes");
  std::memcpy(id128, &uid, sizeof uid);
  return PGP_OK;
}

int pgp_multi_create_ranked(pgp_multi** out, const int* device_ids, int n_local, int rank0, int world, const void* id128) {
  if (n_local <= 0 || rank0 < 0) {
    set_error("pgp_multi_create_ranked: %d local members from rank %d", n_local, rank0);
    return PGP_EINVAL;
  }
  return create_group(out, device_ids, n_local, rank0, world, id128);
}

int pgp_multi_get_info(pgp_multi* m, pgp_multi_info* info) {
  if (!m || !info) {
    set_error("pgp_multi_get_info: bad argument");
    return PGP_EINVAL;
  }
  std::memset(info, 0, sizeof *info);
  info->n_local = m->n;
  info->world = m->world;
  info->rank0 = m->rank0;
  info->emulated = (m->emulate || m->emulate_ranked) ? 1 : 0;
  info->exchanges = m->n_exchanges;
  for (int k = 0; k < m->n && k < 16; ++k) info->devices[k] = m->dev[(size_t)k];
  if (!m->comm.empty() && m->comm[0]) {
    int cnt = 0;
    const ncclResult_t nr = m->rccl.CommCount(m->comm[0], &cnt);
    if (nr != ncclSuccess) {
      set_error("ncclCommCount failed: %s", m->rccl.GetErrorString(nr));
      return PGP_EHIP;
    }
    info->rccl_ranks = cnt;
  }
  return PGP_OK;
}

int pgp_multi_destroy(pgp_multi* m) {
  if (!m) return PGP_OK;
  const bool workers = !m->worker.empty() && (int)m->worker.size() == m->n;
  if (workers) {
    run_all(m, [m](int k) -> int {
      if (m->stream[k]) (void)hipStreamSynchronize(m->stream[k]);
      if (k < (int)m->s2.size() && m->s2[(size_t)k].x) (void)hipStreamSynchronize(m->s2[(size_t)k].x);
      return PGP_OK;
    });
  }
  for (ncclComm_t c : m->comm)
    if (c) m->rccl.CommDestroy(c);
  if (workers) {
    run_all(m, [m](int k) -> int {
      m->d_T[k].release();
      m->d_all[k].release();
      m->d_best[k].release();
      if (k < (int)m->ev.size() && m->ev[k]) (void)hipEventDestroy(m->ev[k]);
      if (k == 0) {
        m->d_sum.release();
        m->d_best2.release();
        m->d_tail_seq.release();
        if (m->up_stream) {
          (void)hipStreamSynchronize(m->up_stream);
          (void)hipStreamDestroy(m->up_stream);
        }
        if (m->ev_rest) (void)hipEventDestroy(m->ev_rest);
        for (hipEvent_t ev : m->ev_sum)
          if (ev) (void)hipEventDestroy(ev);
      }
      if (k < (int)m->s2.size()) {
        Streaming& z = m->s2[(size_t)k];
        if (z.x) (void)hipStreamSynchronize(z.x);
        for (int b = 0; b < 2; ++b) {
          z.ring[b].release();
          if (z.scored[b]) (void)hipEventDestroy(z.scored[b]);
          if (z.reduced[b]) (void)hipEventDestroy(z.reduced[b]);
        }
        if (z.x) (void)hipStreamDestroy(z.x);
      }
      for (auto& per_slot : m->slot_T)
        if (k < (int)per_slot.size()) per_slot[(size_t)k].release();
      if (m->stream[k]) (void)hipStreamDestroy(m->stream[k]);
      for (auto& per_obj : m->octx)
        if (per_obj[(size_t)k]) pgp_destroy(per_obj[(size_t)k]);
      for (pgp_ctx* c : m->ictx[(size_t)k])
        if (c) pgp_destroy(c);
      return PGP_OK;
    });
  }
  for (Worker* w : m->worker) {
    w->shut_down();
    delete w;
  }
  if (m->shm.h) {
    (void)munmap(m->shm.h, m->shm.total);
    if (m->rank0 == 0) (void)shm_unlink(m->shm.name.c_str());
  }
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  // the RCCL handle stays loaded for the life of the process (its own teardown runs at exit)
  delete m;
  return PGP_OK;
}

int pgp_multi_size(const pgp_multi* m) { return m ? m->n : 0; }

pgp_ctx* pgp_multi_context(pgp_multi* m, int k) { return (m && k >= 0 && k < m->n) ? m->octx[0][(size_t)k] : nullptr; }

int pgp_multi_add_object(pgp_multi* m) {
  if (!m) {
    set_error("pgp_multi_add_object: handle is NULL");
    return PGP_EINVAL;
  }
  const int o = m->objects();
  m->octx.emplace_back((size_t)m->n, nullptr);
  m->cs_lo.emplace_back();
  const int rc = run_all(m, [m, o](int k) -> int { return pgp_create(&m->octx[(size_t)o][(size_t)k], m->dev[k]); });
  if (rc != PGP_OK) {
    run_all(m, [m, o](int k) -> int {
      if (m->octx[(size_t)o][(size_t)k]) pgp_destroy(m->octx[(size_t)o][(size_t)k]);
      return PGP_OK;
    });
    m->octx.pop_back();
    m->cs_lo.pop_back();
    return rc;
  }
  return o;
}

int pgp_multi_objects(const pgp_multi* m) { return m ? m->objects() : 0; }

pgp_ctx* pgp_multi_object_context(pgp_multi* m, int obj, int k) {
  return (m && obj >= 0 && obj < m->objects() && k >= 0 && k < m->n) ? m->octx[(size_t)obj][(size_t)k] : nullptr;
}

int pgp_multi_set_object_scene(pgp_multi* m, int obj, const float* xyz, const float* nrm, const float* weight, int n,
                               float delta) {
  if (bad_object(m, obj, "pgp_multi_set_object_scene")) return PGP_EINVAL;
  return run_all(m, [=](int k) -> int { return pgp_set_scene(m->octx[(size_t)obj][(size_t)k], xyz, nrm, weight, n, delta); });
}

int pgp_multi_set_object_scene_weights(pgp_multi* m, int obj, const float* weight, int n) {
  if (bad_object(m, obj, "pgp_multi_set_object_scene_weights")) return PGP_EINVAL;
  return run_all(m, [=](int k) -> int { return pgp_set_scene_weights(m->octx[(size_t)obj][(size_t)k], weight, n); });
}

int pgp_multi_set_object_model(pgp_multi* m, int obj, const float* xyz, const float* nrm, int n) {
  if (bad_object(m, obj, "pgp_multi_set_object_model")) return PGP_EINVAL;
  return run_all(m, [=](int k) -> int { return pgp_set_model(m->octx[(size_t)obj][(size_t)k], xyz, nrm, n); });
}

int pgp_multi_set_object_search_model(pgp_multi* m, int obj, const float* xyz, int n) {
  if (bad_object(m, obj, "pgp_multi_set_object_search_model")) return PGP_EINVAL;
  m->cs_lo[(size_t)obj].clear();
  return run_all(m, [=](int k) -> int { return pgp_set_search_model(m->octx[(size_t)obj][(size_t)k], xyz, n); });
}

int pgp_multi_set_object_ppf_map(pgp_multi* m, int obj, const int* keys, const int* counts, const int* pairs, int n_keys) {
  if (bad_object(m, obj, "pgp_multi_set_object_ppf_map")) return PGP_EINVAL;
  m->cs_lo[(size_t)obj].clear();
  return run_all(m, [=](int k) -> int { return pgp_set_ppf_map(m->octx[(size_t)obj][(size_t)k], keys, counts, pairs, n_keys); });
}

int pgp_multi_set_scene(pgp_multi* m, const float* xyz, const float* nrm, const float* weight, int n, float delta) {
  return pgp_multi_set_object_scene(m, 0, xyz, nrm, weight, n, delta);
}

int pgp_multi_set_scene_weights(pgp_multi* m, const float* weight, int n) {
  return pgp_multi_set_object_scene_weights(m, 0, weight, n);
}

int pgp_multi_set_model(pgp_multi* m, const float* xyz, const float* nrm, int n) {
  return pgp_multi_set_object_model(m, 0, xyz, nrm, n);
}

int pgp_multi_upload(pgp_multi* m, const float* T, int n_h) {
  if (!m || n_h < 0 || (n_h > 0 && !T)) {
    set_error("pgp_multi_upload: bad argument");
    return PGP_EINVAL;
  }
  return upload_flat(m, &T, &n_h, 1);
}

int pgp_multi_upload_objects(pgp_multi* m, const float* const* T, const int* n_h, int n_obj) {
  if (!m || n_obj < 1 || n_obj > m->objects() || !T || !n_h) {
    set_error("pgp_multi_upload_objects: bad argument (%d lists for %d objects)", n_obj, m ? m->objects() : 0);
    return PGP_EINVAL;
  }
  return upload_flat(m, T, n_h, n_obj);
}

int pgp_multi_score_uploaded(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts,
                             int* best_index, float* best_score) {
  if (!m) {
    set_error("pgp_multi_score_uploaded: handle is NULL");
    return PGP_EINVAL;
  }
  if (m->off.size() != 2) {
    set_error("pgp_multi_score_uploaded: the resident batch belongs to %d objects (pgp_multi_score_objects_uploaded)",
              (int)m->off.size() - 1);
    return PGP_ESTATE;
  }
  return score_flat(m, mode, gate_deg, scores, counts, best_index, best_score);
}

int pgp_multi_score_objects_uploaded(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts, int* best_index,
                                     float* best_score) {
  if (!m) {
    set_error("pgp_multi_score_objects_uploaded: handle is NULL");
    return PGP_EINVAL;
  }
  return score_flat(m, mode, gate_deg, scores, counts, best_index, best_score);
}

int pgp_multi_score_lcp(pgp_multi* m, const float* T, int n_h, int mode, float gate_deg, float* scores,
                        int* counts, int* best_index, float* best_score) {
  if (!m || n_h < 0 || (n_h > 0 && (!T || !scores))) {
    set_error("pgp_multi_score_lcp: bad argument");
    return PGP_EINVAL;
  }
  const double t0 = now_ms();
  int rc = pgp_multi_upload(m, T, n_h);
  if (rc != PGP_OK) return rc;
  const float up = (float)(now_ms() - t0);
  rc = pgp_multi_score_uploaded(m, mode, gate_deg, scores, counts, best_index, best_score);
  m->last_ms[0] = up;
  return rc;
}

int pgp_multi_score_objects(pgp_multi* m, const float* const* T, const int* n_h, int n_obj, int mode, float gate_deg,
                            float* scores, int* counts, int* best_index, float* best_score) {
  if (!m || n_obj < 1 || !T || !n_h) {
    set_error("pgp_multi_score_objects: bad argument");
    return PGP_EINVAL;
  }
  const double t0 = now_ms();
  int rc = pgp_multi_upload_objects(m, T, n_h, n_obj);
  if (rc != PGP_OK) return rc;
  if (m->total() > 0 && !scores) {
    set_error("pgp_multi_score_objects: scores is NULL");
    return PGP_EINVAL;
  }
  const float up = (float)(now_ms() - t0);
  rc = score_flat(m, mode, gate_deg, scores, counts, best_index, best_score);
  m->last_ms[0] = up;
  return rc;
}

// ---- streaming form: resident batches, steps queued without a host wait, the exchange under the next step's scoring ----
namespace {
struct SumPtrs {
  const float* p[kEmulateMax];
};
// emulate_sum with the members' vectors passed by value (nothing to upload, nothing to wait for on the host)
__global__ __launch_bounds__(256) void emulate_sum_args(SumPtrs in, int n_members, int n_h, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * n_h) return;
  int s = 0;
  for (int k = 0; k < n_members; ++k) s += reinterpret_cast<const int*>(in.p[k])[i];
  reinterpret_cast<int*>(out)[i] = s;
}

int streaming_ready(pgp_multi* m, int k) {
  Streaming& z = m->s2[(size_t)k];
  if (z.x) return PGP_OK;
  PGP_HIP(hipStreamCreateWithFlags(&z.x, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    PGP_HIP(hipEventCreateWithFlags(&z.scored[b], hipEventDisableTiming));
    PGP_HIP(hipEventCreateWithFlags(&z.reduced[b], hipEventDisableTiming));
    if (k == 0 && m->emulate) PGP_HIP(hipEventCreateWithFlags(&m->ev_sum[b], hipEventDisableTiming));
  }
  if (k == 0) {
    const int r = m->d_best2.ensure(64);
    if (r != PGP_OK) return r;
  }
  return PGP_OK;
}

// member 0, on its EXCHANGE stream behind the all-reduce of ring[b]: the arg-max over the complete vector with the near-tie
// settlement (and the opt-in passes over the complete vector), exactly score_flat's tail -- beside the next step's scoring on the
// member's other stream (a settle over 32 768 scores is ~15 us: an eighth of a step that the scoring stream does not wait for).
// The settlement's re-score workspace is the tail's own (finalize_scores on the scoring stream uses the context's); the opt-in
// passes' workspaces are not touched by the slices' scoring (score_slice switches those passes off).
int queue_tail(pgp_multi* m, int b) {
  Streaming& z = m->s2[0];
  hipStream_t st = z.x;
  const int N = m->pend_n[b], mode = m->pend_mode[b];
  const float gate = m->pend_gate[b];
  if (N <= 0) return PGP_OK;   // (the empty batch: its {-1, 0} was published by the scoring call)
  pgp_ctx* c = m->octx[0][0];
  float* d_s = z.ring[b].as<float>();
  const float* d_T = m->slot_T[(size_t)m->pend_slot[b]][0].as<float>();
  int r = launch_settle_best(c, d_T, N, mode, gate, d_s, m->d_best2.as<int>() + 2 * b, st, m->d_tail_seq.as<float>());
  c->device_work_pending = true;   // (what pgp_settle_best_device's guard notes: work queued outside the context's stream)
  if (r != PGP_OK) return r;
  if (c->exact_records && (r = pgp_settle_records_device(c, d_T, N, mode, gate, d_s, st)) != PGP_OK) return r;
  if (c->verify_early_out && mode == PGP_MODE_PLAIN &&
      (r = pgp_verify_early_out_device(c, d_T, N, d_s, reinterpret_cast<int*>(d_s + N), st)) != PGP_OK)
    return r;
  return PGP_OK;
}
}  // namespace

int pgp_multi_upload_slot(pgp_multi* m, int slot, const float* T, int n_h) {
  if (!m || slot < 0 || slot >= kSlots || n_h < 0 || (n_h > 0 && !T) || n_h > 0x3FFFFFFF) {
    set_error("pgp_multi_upload_slot: bad argument (slot %d of %d, %d hypotheses)", slot, kSlots, n_h);
    return PGP_EINVAL;
  }
  if (m->step != 0) {
    set_error("pgp_multi_upload_slot: %lld steps are in flight (pgp_multi_collect first)", m->step);
    return PGP_ESTATE;
  }
  if (m->emulate && m->n > kEmulateMax) {
    set_error("pgp_multi_upload_slot: the streaming form emulates at most %d members", kEmulateMax);
    return PGP_EINVAL;
  }
  const int N = n_h;
  const size_t nT = (size_t)N * 64;
  int rc = ensure_pin(m, nT + (size_t)N * 8 + 256);
  if (rc != PGP_OK) return rc;
  if (nT) std::memcpy(m->h_pin, T, nT);
  rc = run_all(m, [m, slot, N, nT](int k) -> int {
    int lo, hi, r;
    slice_of(N, m->rank0 + k, m->world, &lo, &hi);
    if ((r = streaming_ready(m, k)) != PGP_OK) return r;
    Streaming& z = m->s2[(size_t)k];
    DevBuf& dT = m->slot_T[(size_t)slot][(size_t)k];
    if ((r = dT.ensure(nT + 64)) != PGP_OK) return r;
    for (int b = 0; b < 2; ++b)
      if ((r = z.ring[b].ensure((size_t)N * 8 + 64)) != PGP_OK) return r;
    if (k == 0 && m->emulate && (r = m->d_sum.ensure((size_t)N * 8 + 64)) != PGP_OK) return r;
    if (k == 0 && (r = m->d_tail_seq.ensure(std::max<size_t>(m->octx[0][0]->d_seq.cap, 64))) != PGP_OK) return r;
    // member 0 settles over the complete vector: its context takes all N, the others their slice
    if ((r = pgp_reserve(m->octx[0][(size_t)k], k == 0 ? N : hi - lo)) != PGP_OK) return r;
    const size_t a = k == 0 ? 0 : (size_t)lo * 64, b = k == 0 ? nT : (size_t)hi * 64;
    if (b > a)
      PGP_HIP(hipMemcpyAsync(dT.as<unsigned char>() + a, static_cast<unsigned char*>(m->h_pin) + a, b - a,
                             hipMemcpyHostToDevice, m->stream[k]));
    PGP_HIP(hipStreamSynchronize(m->stream[k]));   // the pinned image is free again when this call returns
    return PGP_OK;
  });
  m->slot_n[(size_t)slot] = rc == PGP_OK ? N : -1;
  return rc;
}

int pgp_multi_enqueue_slot(pgp_multi* m, int slot, int mode, float gate_deg) {
  if (!m || slot < 0 || slot >= kSlots || m->slot_n[(size_t)slot] < 0) {
    set_error("pgp_multi_enqueue_slot: slot %d holds no batch (pgp_multi_upload_slot)", slot);
    return PGP_EINVAL;
  }
  const int N = m->slot_n[(size_t)slot];
  const int b = (int)(m->step & 1);
  const bool exchange = m->exchange() && N > 0;
  std::vector<int> fail((size_t)m->n, 0);
  m->pend_slot[b] = slot;
  m->pend_mode[b] = mode;
  m->pend_gate[b] = gate_deg;
  m->pend_n[b] = N;
  if (exchange) ++m->n_exchanges;
  const long long step = m->step;
  const int rc = run_all(m, [&, m](int k) -> int {
    Streaming& z = m->s2[(size_t)k];
    hipStream_t S = m->stream[k], X = z.x;
    int lo, hi;
    slice_of(N, m->rank0 + k, m->world, &lo, &hi);
    float* d_s = z.ring[b].as<float>();
    int* d_c = reinterpret_cast<int*>(d_s + N);
    pgp_ctx* c = m->octx[0][(size_t)k];
    // ---- this member's slice into ring[b] (zeroed first: the sum over the members is the gather) ----
    int r1 = [&]() -> int {
      if (!m->exchange())   // one member, no collective: the scoring call publishes the best itself (and runs the opt-in passes)
        return pgp_score_lcp_device(c, m->slot_T[(size_t)slot][(size_t)k].as<float>(), N, mode, gate_deg, d_s, d_c,
                                    m->d_best2.as<int>() + 2 * b, S);
      // ring[b] was last used two steps ago: its exchange (and, on member 0, its tail behind it) must be through
      if (step >= 2) PGP_HIP(hipStreamWaitEvent(S, z.reduced[b], 0));
      if (N > 0 && m->world > 1) PGP_HIP(hipMemsetAsync(d_s, 0, (size_t)N * 8, S));
      const bool records = c->exact_records, early = c->verify_early_out;
      c->exact_records = false;   // they belong to the complete vector (queue_tail)
      c->verify_early_out = false;
      const int r = hi > lo || N == 0
                        ? pgp_score_lcp_device(c, m->slot_T[(size_t)slot][(size_t)k].as<float>() + 16 * (size_t)lo, hi - lo, mode,
                                               gate_deg, d_s + lo, d_c + lo, N == 0 && k == 0 ? m->d_best2.as<int>() + 2 * b : nullptr, S)
                        : PGP_OK;
      c->exact_records = records;
      c->verify_early_out = early;
      return r;
    }();
    if (!exchange) return r1;   // (one member without a collective, or the empty batch: nothing to exchange, nothing to settle)
    // ---- the exchange of ring[b] on the second stream ----
    if (r1 == PGP_OK && hipEventRecord(z.scored[b], S) != hipSuccess) r1 = PGP_EHIP;
    if (r1 == PGP_OK && hipStreamWaitEvent(X, z.scored[b], 0) != hipSuccess) r1 = PGP_EHIP;
    int r2 = PGP_OK;
    if (m->emulate) {
      fail[(size_t)k] = r1 != PGP_OK;
      m->barrier();   // every member's slice is queued (or has failed)
      bool any = false;
      for (int f : fail) any = any || f;
      if (k == 0 && !any) {
        r2 = [&]() -> int {
          SumPtrs ptrs{};
          for (int j = 0; j < m->n; ++j) {
            ptrs.p[j] = m->s2[(size_t)j].ring[b].as<float>();
            if (j > 0) PGP_HIP(hipStreamWaitEvent(X, m->s2[(size_t)j].scored[b], 0));
          }
          float* d_out = m->d_sum.as<float>();
          hipLaunchKernelGGL(emulate_sum_args, dim3((2 * N + 255) / 256), dim3(256), 0, X, ptrs, m->n, N, d_out);
          PGP_HIP(hipGetLastError());
          for (int j = 0; j < m->n; ++j)
            PGP_HIP(hipMemcpyAsync(m->s2[(size_t)j].ring[b].p, d_out, (size_t)N * 8, hipMemcpyDeviceToDevice, X));
          PGP_HIP(hipEventRecord(m->ev_sum[b], X));
          return PGP_OK;
        }();
        fail[0] = r2 != PGP_OK;
      }
      m->barrier();   // the sum is queued
      if (r1 == PGP_OK && r2 == PGP_OK && (any || fail[0])) {
        set_error("another member of the group failed");
        r2 = PGP_EHIP;
      }
      if (r1 == PGP_OK && r2 == PGP_OK && k > 0) PGP_HIP(hipStreamWaitEvent(X, m->ev_sum[b], 0));
    } else {
      // (a member whose slice failed to queue still joins the collective: the others' streams would wait for it forever)
      if (m->emulate_ranked) {
        r2 = shm_exchange(m, d_s, N, X);   // (host-synchronous: X has waited for the slice)
      } else {
        const ncclResult_t nr = m->rccl.AllReduce(d_s, d_s, 2 * (size_t)N, ncclInt32, ncclSum, m->comm[(size_t)k], X);
        if (nr != ncclSuccess) {
          set_error("ncclAllReduce failed: %s", m->rccl.GetErrorString(nr));
          r2 = PGP_EHIP;
        }
      }
    }
    if (r1 != PGP_OK) return r1;
    if (r2 != PGP_OK) return r2;
    // member 0: the step's tail behind its exchange, on the exchange stream -- beside the next step's scoring
    if (k == 0) {
      if (m->d_tail_seq.cap < c->d_seq.cap) {   // (the model changed since the batches went up)
        const int r = m->d_tail_seq.ensure(c->d_seq.cap);
        if (r != PGP_OK) return r;
      }
      const int r = queue_tail(m, b);
      if (r != PGP_OK) return r;
    }
    PGP_HIP(hipEventRecord(z.reduced[b], X));   // ring[b] is free again (two steps on) once this has passed
    return PGP_OK;
  });
  if (rc != PGP_OK) return rc;
  ++m->step;
  return PGP_OK;
}

int pgp_multi_collect(pgp_multi* m, float* scores, int* counts, int* best_index, float* best_score) {
  if (!m) {
    set_error("pgp_multi_collect: handle is NULL");
    return PGP_EINVAL;
  }
  if (m->step == 0) {
    set_error("pgp_multi_collect: no step has been enqueued (pgp_multi_enqueue_slot)");
    return PGP_ESTATE;
  }
  const int b = (int)((m->step - 1) & 1);
  const int N = m->pend_n[b];
  unsigned char* pin_out = static_cast<unsigned char*>(m->h_pin);   // (sized by pgp_multi_upload_slot; no upload is in flight)
  const int rc = run_all(m, [&, m](int k) -> int {
    Streaming& z = m->s2[(size_t)k];
    hipStream_t S = m->stream[k];
    PGP_HIP(hipStreamSynchronize(S));
    if (k == 0) {
      // the last step's arrays: behind its tail on the exchange stream (no exchange: behind the scoring call, just waited for)
      hipStream_t cs = m->exchange() && z.x ? z.x : S;
      if (N > 0) PGP_HIP(hipMemcpyAsync(pin_out, z.ring[b].p, (size_t)N * 8, hipMemcpyDeviceToHost, cs));
      PGP_HIP(hipMemcpyAsync(pin_out + (size_t)N * 8, m->d_best2.as<int>() + 2 * b, 8, hipMemcpyDeviceToHost, cs));
      if (cs == S) PGP_HIP(hipStreamSynchronize(S));
    }
    if (z.x) PGP_HIP(hipStreamSynchronize(z.x));
    return PGP_OK;
  });
  m->step = 0;
  if (rc != PGP_OK) return rc;
  if (N > 0) {
    if (scores) std::memcpy(scores, pin_out, (size_t)N * 4);
    if (counts) std::memcpy(counts, pin_out + (size_t)N * 4, (size_t)N * 4);
  }
  int best[2];
  std::memcpy(best, pin_out + (size_t)N * 8, sizeof best);
  if (best_index) *best_index = best[0];
  if (best_score) std::memcpy(best_score, &best[1], 4);
  return PGP_OK;
}

int pgp_multi_last_timing(pgp_multi* m, float* upload_ms, float* enqueue_ms, float* total_ms) {
  if (!m) {
    set_error("pgp_multi_last_timing: handle is NULL");
    return PGP_EINVAL;
  }
  if (upload_ms) *upload_ms = m->last_ms[0];
  if (enqueue_ms) *enqueue_ms = m->last_ms[1];
  if (total_ms) *total_ms = m->last_ms[2];
  return PGP_OK;
}

// ---- ICP: the poses to refine, sharded like the hypotheses (UCTSearch.cpp:200-266 -> UCTState.cpp:121-204) ----
int pgp_multi_icp_refine(pgp_multi* m, const pgp_multi_icp_job* jobs, int n_jobs, const pgp_icp_params* params) {
  if (!m || n_jobs < 0 || (n_jobs > 0 && !jobs) || !params) {
    set_error("pgp_multi_icp_refine: bad argument");
    return PGP_EINVAL;
  }
  std::vector<int> cnt((size_t)n_jobs, 0);
  for (int j = 0; j < n_jobs; ++j) {
    const pgp_multi_icp_job& q = jobs[j];
    if (q.n < 0 || q.n_src < 0 || q.n_tgt < 0 || (q.n > 0 && (!q.T || !q.src_xyz || !q.tgt_xyz || q.n_src == 0 || q.n_tgt == 0))) {
      set_error("pgp_multi_icp_refine: bad job %d", j);
      return PGP_EINVAL;
    }
    cnt[(size_t)j] = q.n;
  }
  if (!single_process(m, "pgp_multi_icp_refine")) return PGP_ESTATE;
  if (n_jobs == 0) return PGP_OK;
  const pgp_icp_options opt = icp_options_of(params);
  return run_all(m, [&, m](int k) -> int {
    // this member's share of the flat (job, pose) space: pieces (job, lo, hi)
    std::vector<int> pj((size_t)n_jobs), plo((size_t)n_jobs), phi((size_t)n_jobs);
    int np = 0, r;
    if ((r = pgp_multi_flat_slices(cnt.data(), n_jobs, k, m->n, pj.data(), plo.data(), phi.data(), &np)) != PGP_OK) return r;
    if (np == 0) return PGP_OK;
    hipStream_t st = m->stream[k];
    std::vector<pgp_ctx*>& pool = m->ictx[(size_t)k];
    // Every piece needs a context of its own (a context stages ONE job and keeps ONE target index).  Up to kIcpSlots jobs,
    // job j uses the member's context j, so that its target's index stays resident from call to call; beyond that the pieces
    // take the contexts in order (a target is then recognised by its hash, and rebuilt when the slot held another one).
    constexpr int kIcpSlots = 32;
    const bool by_job = n_jobs <= kIcpSlots;
    const int want = by_job ? n_jobs : std::min(np, n_jobs);
    while ((int)pool.size() < want) {
      pgp_ctx* c = nullptr;
      if ((r = pgp_create(&c, m->dev[k])) != PGP_OK) return r;
      pool.push_back(c);
    }
    std::vector<IcpHostStage> stage((size_t)np);
    std::vector<IcpJob> dj((size_t)np);
    for (int p = 0; p < np; ++p) {
      const pgp_multi_icp_job& q = jobs[pj[(size_t)p]];
      pgp_ctx* c = pool[(size_t)(by_job ? pj[(size_t)p] : p)];
      const int n = phi[(size_t)p] - plo[(size_t)p];
      if ((r = icp_host_stage(c, q.src_xyz, q.n_src, q.tgt_xyz, q.n_tgt, q.T + 16 * (size_t)plo[(size_t)p], n, st, &stage[(size_t)p])) != PGP_OK)
        return r;
      const IcpHostStage& g = stage[(size_t)p];
      dj[(size_t)p] = IcpJob{c, g.d_src, q.n_src, g.d_tgt, q.n_tgt, g.d_T, n, g.d_energy, g.d_iters, g.token};
    }
    // ONE launch for the member's pieces (pgp_icp_refine_multi_device; a single piece is the plain call)
    if (np == 1)
      r = launch_icp(dj[0].ctx, dj[0].d_src, dj[0].n_src, dj[0].d_tgt, nullptr, dj[0].n_tgt, dj[0].d_T, dj[0].n, &opt,
                     dj[0].d_energy, dj[0].d_iters, st, dj[0].token);
    else
      r = launch_icp_multi(dj.data(), np, &opt, st);
    if (r != PGP_OK) return r;
    for (int p = 0; p < np; ++p)
      if ((r = icp_host_collect_enqueue(dj[(size_t)p].ctx, stage[(size_t)p], st)) != PGP_OK) return r;
    PGP_HIP(hipStreamSynchronize(st));
    // A piece whose pose the one-launch scene-sized form gave up on (iteration count -1, transform left where it was:
    // icp.hip icp_scene_persist; pgp.h promises host-pointer entries redo such a job) -- that piece once more through
    // the host-driven form, exactly as pgp_icp_refine_ex does, so that the caller sees one pgp_icp_refine per job.
    for (int p = 0; p < np; ++p) {
      const IcpHostStage& g = stage[(size_t)p];
      pgp_ctx* c = dj[(size_t)p].ctx;
      const int* it = reinterpret_cast<const int*>(static_cast<const unsigned char*>(c->h_pin) + g.off_i);
      bool lost = false;
      for (int i = 0; i < dj[(size_t)p].n; ++i) lost = lost || it[i] < 0;
      if (!lost) continue;
      struct Off {
        Off() { pgp::icp_scene_form_off(true); }
        ~Off() { pgp::icp_scene_form_off(false); }
      } off;
      const pgp_multi_icp_job& q = jobs[pj[(size_t)p]];
      const int n = dj[(size_t)p].n;
      if ((r = icp_host_stage(c, q.src_xyz, q.n_src, q.tgt_xyz, q.n_tgt, q.T + 16 * (size_t)plo[(size_t)p], n, st, &stage[(size_t)p])) != PGP_OK)
        return r;
      const IcpHostStage& g2 = stage[(size_t)p];
      dj[(size_t)p] = IcpJob{c, g2.d_src, q.n_src, g2.d_tgt, q.n_tgt, g2.d_T, n, g2.d_energy, g2.d_iters, g2.token};
      if ((r = launch_icp(c, g2.d_src, q.n_src, g2.d_tgt, nullptr, q.n_tgt, g2.d_T, n, &opt, g2.d_energy, g2.d_iters, st, g2.token)) != PGP_OK)
        return r;
      if ((r = icp_host_collect_enqueue(c, g2, st)) != PGP_OK) return r;
      PGP_HIP(hipStreamSynchronize(st));
    }
    // the gather: every member writes its poses' results into the caller's arrays (disjoint ranges, no collective)
    for (int p = 0; p < np; ++p) {
      const pgp_multi_icp_job& q = jobs[pj[(size_t)p]];
      const size_t lo = (size_t)plo[(size_t)p];
      icp_host_collect(dj[(size_t)p].ctx, stage[(size_t)p], dj[(size_t)p].n, q.T + 16 * lo, q.energy ? q.energy + lo : nullptr,
                       q.iters ? q.iters + lo : nullptr);
    }
    return PGP_OK;
  });
}

// ---- congruent sets: the bases of an object sharded over the members (base.cc:1855-1874) ----
int pgp_multi_find_congruent_batch(pgp_multi* m, int obj, const int* base_ids, const float* base_xyz, const float* invariants,
                                   int n_bases, float threshold, int* n_quads) {
  if (bad_object(m, obj, "pgp_multi_find_congruent_batch")) return PGP_EINVAL;
  if (!single_process(m, "pgp_multi_find_congruent_batch")) return PGP_ESTATE;
  if (n_bases < 0 || (n_bases > 0 && (!base_ids || !base_xyz || !invariants || !n_quads))) {
    set_error("pgp_multi_find_congruent_batch: bad argument");
    return PGP_EINVAL;
  }
  std::vector<int>& lo = m->cs_lo[(size_t)obj];
  lo.assign((size_t)m->n + 1, 0);
  for (int k = 0; k < m->n; ++k) {
    int a, b;
    slice_of(n_bases, k, m->n, &a, &b);
    lo[(size_t)k] = a;
    lo[(size_t)k + 1] = b;
  }
  const int rc = run_all(m, [=, &lo](int k) -> int {
    const int a = lo[(size_t)k], b = lo[(size_t)k + 1];
    // (a member without bases still runs the call: it leaves an empty batch behind, as a single context would)
    return pgp_find_congruent_batch(m->octx[(size_t)obj][(size_t)k], base_ids + 4 * (size_t)a, base_xyz + 12 * (size_t)a,
                                    invariants + 2 * (size_t)a, b - a, threshold, n_quads + a);
  });
  if (rc != PGP_OK) lo.clear();
  return rc;
}

namespace {
// picks (base, j) -> the member that owns the base, with the base renumbered inside that member's batch
int route_picks(pgp_multi* m, int obj, const int* picks, int cnt, const char* who, std::vector<std::vector<int>>* where,
                std::vector<std::vector<int>>* local) {
  const std::vector<int>& lo = m->cs_lo[(size_t)obj];
  if (lo.size() != (size_t)m->n + 1) {
    set_error("%s: no congruent batch resident for object %d (pgp_multi_find_congruent_batch)", who, obj);
    return PGP_ESTATE;
  }
  where->assign((size_t)m->n, {});
  local->assign((size_t)m->n, {});
  for (int i = 0; i < cnt; ++i) {
    const int b = picks[2 * (size_t)i];
    if (b < 0 || b >= lo[(size_t)m->n]) {
      set_error("%s: pick %d names base %d of %d", who, i, b, lo[(size_t)m->n]);
      return PGP_EINVAL;
    }
    int k = 0;
    while (b >= lo[(size_t)k + 1]) ++k;
    (*where)[(size_t)k].push_back(i);
    (*local)[(size_t)k].push_back(b - lo[(size_t)k]);
    (*local)[(size_t)k].push_back(picks[2 * (size_t)i + 1]);
  }
  return PGP_OK;
}
}  // namespace

int pgp_multi_congruent_batch_quads(pgp_multi* m, int obj, const int* picks, int cnt, int* quads) {
  if (bad_object(m, obj, "pgp_multi_congruent_batch_quads")) return PGP_EINVAL;
  if (cnt < 0 || (cnt > 0 && (!picks || !quads))) {
    set_error("pgp_multi_congruent_batch_quads: bad argument");
    return PGP_EINVAL;
  }
  if (cnt == 0) return PGP_OK;
  std::vector<std::vector<int>> where, local;
  int rc = route_picks(m, obj, picks, cnt, "pgp_multi_congruent_batch_quads", &where, &local);
  if (rc != PGP_OK) return rc;
  return run_all(m, [&, m, obj, quads](int k) -> int {
    const int c = (int)where[(size_t)k].size();
    if (c == 0) return PGP_OK;
    std::vector<int> q((size_t)c * 4);
    const int r = pgp_congruent_batch_quads(m->octx[(size_t)obj][(size_t)k], local[(size_t)k].data(), c, q.data());
    if (r != PGP_OK) return r;
    for (int i = 0; i < c; ++i) std::memcpy(quads + 4 * (size_t)where[(size_t)k][(size_t)i], &q[(size_t)i * 4], 16);
    return PGP_OK;
  });
}

int pgp_multi_congruent_batch_fit(pgp_multi* m, int obj, const int* picks, const int* base_ids, int cnt,
                                  const float centroid_P[3], const float centroid_Q[3], float* T, double* pose, int* status,
                                  float* rms) {
  if (bad_object(m, obj, "pgp_multi_congruent_batch_fit")) return PGP_EINVAL;
  if (cnt < 0 || (cnt > 0 && (!picks || !base_ids || !T || !status)) || !centroid_P || !centroid_Q) {
    set_error("pgp_multi_congruent_batch_fit: bad argument");
    return PGP_EINVAL;
  }
  if (cnt == 0) return PGP_OK;
  std::vector<std::vector<int>> where, local;
  int rc = route_picks(m, obj, picks, cnt, "pgp_multi_congruent_batch_fit", &where, &local);
  if (rc != PGP_OK) return rc;
  // the variable-length fit lists of the members, gathered on the host in the caller's pick order
  return run_all(m, [&, m, obj](int k) -> int {
    const std::vector<int>& w = where[(size_t)k];
    const int c = (int)w.size();
    if (c == 0) return PGP_OK;
    std::vector<int> st((size_t)c);
    std::vector<float> t((size_t)c * 16), e((size_t)c);
    std::vector<double> ps(pose ? (size_t)c * 16 : 0);
    // base_ids is indexed by base: the member's batch numbers its bases from its first one
    const int* ids = base_ids + 4 * (size_t)m->cs_lo[(size_t)obj][(size_t)k];
    const int r = pgp_congruent_batch_fit(m->octx[(size_t)obj][(size_t)k], local[(size_t)k].data(), ids, c, centroid_P,
                                          centroid_Q, t.data(), pose ? ps.data() : nullptr, st.data(), rms ? e.data() : nullptr);
    if (r != PGP_OK) return r;
    for (int i = 0; i < c; ++i) {
      const size_t d = (size_t)w[(size_t)i];
      std::memcpy(T + 16 * d, &t[(size_t)i * 16], 64);
      if (pose) std::memcpy(pose + 16 * d, &ps[(size_t)i * 16], 128);
      status[d] = st[(size_t)i];
      if (rms) rms[d] = e[(size_t)i];
    }
    return PGP_OK;
  });
}

}  // extern "C"
