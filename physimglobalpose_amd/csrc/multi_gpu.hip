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

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pgp {
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_rccl(Rccl* r) {
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
  PGP_SYM(CommDestroy, "ncclCommDestroy")
  PGP_SYM(AllReduce, "ncclAllReduce")
  PGP_SYM(GroupStart, "ncclGroupStart")
  PGP_SYM(GroupEnd, "ncclGroupEnd")
  PGP_SYM(GetErrorString, "ncclGetErrorString")
#undef PGP_SYM
  return true;
}

// One host thread per device: runs the jobs posted to it with its device current.
struct Worker {
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool has_job = false, stop = false, done = true;
  int rc = PGP_OK;
  char err[512] = "";

  void loop() {
    (void)hipSetDevice(device);
    for (;;) {
      std::function<int()> j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return has_job || stop; });
        if (stop) return;
        j = std::move(job);
        has_job = false;
      }
      int r = j();
      {
        std::lock_guard<std::mutex> lk(mu);
        rc = r;
        if (r != PGP_OK) {
          std::strncpy(err, pgp_last_error(), sizeof err - 1);
          err[sizeof err - 1] = 0;
        }
        done = true;
      }
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has_job = true;
      done = false;
    }
    cv.notify_all();
  }
  int wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return rc;
  }
};

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace
}  // namespace pgp

using namespace pgp;

struct pgp_multi {
  int n = 0;
  std::vector<int> dev;
  std::vector<pgp_ctx*> ctx;
  std::vector<hipStream_t> stream;
  std::vector<Worker*> worker;
  bool use_coll = false;
  bool emulate = false;   // PGP_MULTI_EMULATE: members share one device, exchange = emulate_allreduce
  std::vector<hipEvent_t> ev;   // emulate: one event per member
  DevBuf d_sum;                 // emulate: the summed vector before it is handed to every member
  Rccl rccl;
  std::vector<ncclComm_t> comm;
  // per device: the full transform list and the full-length {scores | counts} vector
  std::vector<DevBuf> d_T, d_all, d_best;
  int n_h = 0;          // hypotheses currently uploaded
  void* h_pin = nullptr;  // portable pinned staging: transforms in, scores | counts | best out
  size_t h_pin_cap = 0;
  float last_ms[3] = {0.f, 0.f, 0.f};  // host wall clock of the last call: upload, enqueue, total
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

int run_all(pgp_multi* m, const std::function<int(int)>& fn) {
  for (int k = 0; k < m->n; ++k) m->worker[k]->post([&fn, k] { return fn(k); });
  int rc = PGP_OK;
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

}  // namespace

extern "C" {

int pgp_multi_slice(int n_total, int k, int n_dev, int* lo, int* hi) {
  if (n_total < 0 || n_dev <= 0 || k < 0 || k >= n_dev || !lo || !hi) {
    set_error("pgp_multi_slice: bad argument");
    return PGP_EINVAL;
  }
  // contiguous slices, sizes differ by at most one, earlier devices larger (sharding.shard_bounds)
  const int base = n_total / n_dev, rem = n_total % n_dev;
  *lo = k * base + (k < rem ? k : rem);
  *hi = *lo + base + (k < rem ? 1 : 0);
  return PGP_OK;
}

int pgp_multi_create(pgp_multi** out, const int* device_ids, int n_dev) {
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
  int emulate = 0;
  if (const char* v = getenv("PGP_MULTI_EMULATE")) emulate = atoi(v);
  if (emulate >= 2) n_dev = emulate;   // n logical members on the first listed device
  if (n_dev <= 0) n_dev = visible;  // every visible device
  pgp_multi* m = new pgp_multi();
  m->n = n_dev;
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
  m->ctx.assign(n_dev, nullptr);
  m->stream.assign(n_dev, nullptr);
  m->d_T.resize(n_dev);
  m->d_all.resize(n_dev);
  m->d_best.resize(n_dev);
  int rc = PGP_OK;
  for (int k = 0; k < n_dev && rc == PGP_OK; ++k) {
    Worker* w = new Worker();
    w->device = m->dev[k];
    w->th = std::thread([w] { w->loop(); });
    m->worker.push_back(w);
  }
  rc = run_all(m, [m](int k) -> int {
    int r = pgp_create(&m->ctx[k], m->dev[k]);
    if (r != PGP_OK) return r;
    PGP_HIP(hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking));
    return m->d_best[k].ensure(16);
  });
  const char* force = getenv("PGP_MULTI_FORCE_COLLECTIVE");
  m->use_coll = n_dev > 1 || (force && atoi(force) != 0);
  if (rc == PGP_OK && m->emulate) {
    m->ev.assign(n_dev, nullptr);
    rc = run_all(m, [m](int k) -> int {
      PGP_HIP(hipEventCreateWithFlags(&m->ev[k], hipEventDisableTiming));
      return PGP_OK;
    });
  }
  if (rc == PGP_OK && m->use_coll && !m->emulate) {
    if (!load_rccl(&m->rccl)) rc = PGP_ENODEV;
    if (rc == PGP_OK) {
      m->comm.assign(n_dev, nullptr);
      ncclResult_t nr = m->rccl.CommInitAll(m->comm.data(), n_dev, m->dev.data());
      if (nr != ncclSuccess) {
        set_error("ncclCommInitAll failed: %s", m->rccl.GetErrorString(nr));
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

int pgp_multi_destroy(pgp_multi* m) {
  if (!m) return PGP_OK;
  if (!m->worker.empty() && (int)m->worker.size() == m->n) {
    run_all(m, [m](int k) -> int {
      if (m->stream[k]) (void)hipStreamSynchronize(m->stream[k]);
      return PGP_OK;
    });
  }
  for (ncclComm_t c : m->comm)
    if (c) m->rccl.CommDestroy(c);
  if (!m->worker.empty() && (int)m->worker.size() == m->n) {
    run_all(m, [m](int k) -> int {
      m->d_T[k].release();
      m->d_all[k].release();
      m->d_best[k].release();
      if (k < (int)m->ev.size() && m->ev[k]) (void)hipEventDestroy(m->ev[k]);
      if (k == 0) m->d_sum.release();
      if (m->stream[k]) (void)hipStreamDestroy(m->stream[k]);
      if (m->ctx[k]) pgp_destroy(m->ctx[k]);
      return PGP_OK;
    });
  }
  for (Worker* w : m->worker) {
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
    }
    w->cv.notify_all();
    if (w->th.joinable()) w->th.join();
    delete w;
  }
  if (m->h_pin) (void)hipHostFree(m->h_pin);
  // the RCCL handle stays loaded for the life of the process (its own teardown runs at exit)
  delete m;
  return PGP_OK;
}

int pgp_multi_size(const pgp_multi* m) { return m ? m->n : 0; }

pgp_ctx* pgp_multi_context(pgp_multi* m, int k) { return (m && k >= 0 && k < m->n) ? m->ctx[k] : nullptr; }

int pgp_multi_set_scene(pgp_multi* m, const float* xyz, const float* nrm, const float* weight, int n, float delta) {
  if (!m) {
    set_error("pgp_multi_set_scene: handle is NULL");
    return PGP_EINVAL;
  }
  return run_all(m, [=](int k) -> int { return pgp_set_scene(m->ctx[k], xyz, nrm, weight, n, delta); });
}

int pgp_multi_set_scene_weights(pgp_multi* m, const float* weight, int n) {
  if (!m) {
    set_error("pgp_multi_set_scene_weights: handle is NULL");
    return PGP_EINVAL;
  }
  return run_all(m, [=](int k) -> int { return pgp_set_scene_weights(m->ctx[k], weight, n); });
}

int pgp_multi_set_model(pgp_multi* m, const float* xyz, const float* nrm, int n) {
  if (!m) {
    set_error("pgp_multi_set_model: handle is NULL");
    return PGP_EINVAL;
  }
  return run_all(m, [=](int k) -> int { return pgp_set_model(m->ctx[k], xyz, nrm, n); });
}

int pgp_multi_upload(pgp_multi* m, const float* T, int n_h) {
  if (!m || n_h < 0 || (n_h > 0 && !T)) {
    set_error("pgp_multi_upload: bad argument");
    return PGP_EINVAL;
  }
  const size_t nT = (size_t)n_h * 64;
  int rc = ensure_pin(m, nT + (size_t)n_h * 8 + 64);
  if (rc != PGP_OK) return rc;
  if (nT) std::memcpy(m->h_pin, T, nT);
  m->n_h = n_h;
  return run_all(m, [m, nT, n_h](int k) -> int {
    int lo, hi;
    pgp_multi_slice(n_h, k, m->n, &lo, &hi);
    int r;
    if ((r = m->d_T[k].ensure(nT)) != PGP_OK) return r;
    if ((r = m->d_all[k].ensure((size_t)n_h * 8)) != PGP_OK) return r;
    // member 0 also works on the complete vector (settlement, exact records, Verify's early termination)
    if ((r = pgp_reserve(m->ctx[k], k == 0 ? n_h : hi - lo)) != PGP_OK) return r;
    // every device holds ALL transforms: device 0 needs them to settle near-ties across slices
    // (64 B per hypothesis: 4 MB at 65 536 -- each device pulls its copy over its own PCIe link)
    if (nT) PGP_HIP(hipMemcpyAsync(m->d_T[k].p, m->h_pin, nT, hipMemcpyHostToDevice, m->stream[k]));
    return PGP_OK;
  });
}

int pgp_multi_score_uploaded(pgp_multi* m, int mode, float gate_deg, float* scores, int* counts,
                             int* best_index, float* best_score) {
  if (!m) {
    set_error("pgp_multi_score_uploaded: handle is NULL");
    return PGP_EINVAL;
  }
  const int n_h = m->n_h;
  const double t0 = now_ms();
  int rc = ensure_pin(m, 64);   // nothing uploaded yet: the empty batch still returns {-1, 0}
  if (rc != PGP_OK) return rc;
  rc = run_all(m, [m, mode, gate_deg, n_h](int k) -> int {
    int lo, hi;
    pgp_multi_slice(n_h, k, m->n, &lo, &hi);
    float* d_s = m->d_all[k].as<float>();
    int* d_c = reinterpret_cast<int*>(d_s + n_h);
    // every device fills only its slice of a zeroed vector: the sum over devices is the gather
    if (n_h > 0 && m->n > 1) PGP_HIP(hipMemsetAsync(d_s, 0, (size_t)n_h * 8, m->stream[k]));
    // the exact-records pass (pgp_set_exact_records on member 0's context) belongs to the COMPLETE vector,
    // below; running it on member 0's slice as well would only repeat three launches
    const bool records = m->ctx[k]->exact_records, early = m->ctx[k]->verify_early_out;
    m->ctx[k]->exact_records = false;
    m->ctx[k]->verify_early_out = false;   // Verify's early termination depends on ALL earlier hypotheses: below
    const int r = pgp_score_lcp_device(m->ctx[k], m->d_T[k].as<float>() + 16 * (size_t)lo, hi - lo, mode, gate_deg,
                                       d_s + lo, d_c + lo, nullptr, m->stream[k]);
    m->ctx[k]->exact_records = records;
    m->ctx[k]->verify_early_out = early;
    if (r == PGP_OK && m->emulate) PGP_HIP(hipEventRecord(m->ev[k], m->stream[k]));
    return r;
  });
  if (rc != PGP_OK) return rc;
  if (m->emulate && m->n > 1 && n_h > 0) {
    // the all-reduce on one device: member 0's stream waits for every slice, sums the vectors, and hands
    // the sum to every member (whose streams then wait for it)
    Worker* w0 = m->worker[0];
    w0->post([m, n_h]() -> int {
      hipStream_t st = m->stream[0];
      int r;
      if ((r = m->d_sum.ensure((size_t)n_h * 8 + (size_t)m->n * sizeof(float*) + 64)) != PGP_OK) return r;
      float* d_out = m->d_sum.as<float>();
      const float** d_ptrs = reinterpret_cast<const float**>(m->d_sum.as<unsigned char>() + (((size_t)n_h * 8 + 15) & ~(size_t)15));
      std::vector<const float*> ptrs(m->n);
      for (int k = 0; k < m->n; ++k) {
        ptrs[k] = m->d_all[k].as<float>();
        if (k > 0) PGP_HIP(hipStreamWaitEvent(st, m->ev[k], 0));
      }
      PGP_HIP(hipMemcpyAsync(d_ptrs, ptrs.data(), (size_t)m->n * sizeof(float*), hipMemcpyHostToDevice, st));
      PGP_HIP(hipStreamSynchronize(st));   // ptrs is a stack temporary
      hipLaunchKernelGGL(emulate_sum, dim3((2 * n_h + 255) / 256), dim3(256), 0, st, d_ptrs, m->n, n_h, d_out);
      PGP_HIP(hipGetLastError());
      for (int k = 0; k < m->n; ++k)
        PGP_HIP(hipMemcpyAsync(m->d_all[k].p, d_out, (size_t)n_h * 8, hipMemcpyDeviceToDevice, st));
      PGP_HIP(hipEventRecord(m->ev[0], st));
      return PGP_OK;
    });
    rc = w0->wait();
    if (rc != PGP_OK) {
      set_error("device %d: %s", m->dev[0], w0->err);
      return rc;
    }
    rc = run_all(m, [m](int k) -> int {
      if (k > 0) PGP_HIP(hipStreamWaitEvent(m->stream[k], m->ev[0], 0));
      return PGP_OK;
    });
    if (rc != PGP_OK) return rc;
  } else if (m->use_coll && n_h > 0) {
    // ONE all-reduce per call over the 8 n_h bytes {scores | counts}, summed as 32-bit integers: every element is
    // non-zero on exactly one member (its owner) and all-zero bits elsewhere, and x + 0 + ... + 0 over the BIT
    // PATTERNS returns x's pattern -- exact for the float scores too (scores are >= +0: no -0, no NaN).  At 4096
    // hypotheses the exchange is latency-bound (32 KB per member), so two collectives cost twice what one does.
    ncclResult_t nr = m->rccl.GroupStart();
    for (int k = 0; k < m->n && nr == ncclSuccess; ++k) {
      float* d_s = m->d_all[k].as<float>();
      nr = m->rccl.AllReduce(d_s, d_s, 2 * (size_t)n_h, ncclInt32, ncclSum, m->comm[k], m->stream[k]);
    }
    ncclResult_t ge = m->rccl.GroupEnd();
    if (nr == ncclSuccess) nr = ge;
    if (nr != ncclSuccess) {
      set_error("ncclAllReduce failed: %s", m->rccl.GetErrorString(nr));
      return PGP_EHIP;
    }
  }
  const double t1 = now_ms();
  // device 0: arg-max over the complete vector (exact under weighted near-ties), one copy back
  unsigned char* pin_out = static_cast<unsigned char*>(m->h_pin) + (((size_t)n_h * 64 + 63) & ~(size_t)63);
  Worker* w0 = m->worker[0];
  w0->post([m, mode, gate_deg, n_h, pin_out]() -> int {
    float* d_s = m->d_all[0].as<float>();
    int* d_b = m->d_best[0].as<int>();
    hipStream_t st = m->stream[0];
    int r = pgp_settle_best_device(m->ctx[0], m->d_T[0].as<float>(), n_h, mode, gate_deg, d_s, d_b, st);
    if (r != PGP_OK) return r;
    // pgp_set_exact_records on device 0's context (pgp_multi_context(m, 0)) covers the group's calls too
    if (m->ctx[0]->exact_records) {
      r = pgp_settle_records_device(m->ctx[0], m->d_T[0].as<float>(), n_h, mode, gate_deg, d_s, st);
      if (r != PGP_OK) return r;
    }
    // pgp_set_verify_early_out on member 0's context: applied to the complete vector of true counts
    if (m->ctx[0]->verify_early_out && mode == PGP_MODE_PLAIN && n_h > 0) {
      r = pgp_verify_early_out_device(m->ctx[0], m->d_T[0].as<float>(), n_h, d_s, reinterpret_cast<int*>(d_s + n_h), st);
      if (r != PGP_OK) return r;
    }
    if (n_h > 0) PGP_HIP(hipMemcpyAsync(pin_out, d_s, (size_t)n_h * 8, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipMemcpyAsync(pin_out + (size_t)n_h * 8, d_b, 8, hipMemcpyDeviceToHost, st));
    PGP_HIP(hipStreamSynchronize(st));
    return PGP_OK;
  });
  rc = w0->wait();
  if (rc != PGP_OK) {
    set_error("device %d: %s", m->dev[0], w0->err);
    return rc;
  }
  if (n_h > 0) {
    if (scores) std::memcpy(scores, pin_out, (size_t)n_h * 4);
    if (counts) std::memcpy(counts, pin_out + (size_t)n_h * 4, (size_t)n_h * 4);
  }
  int best[2];
  std::memcpy(best, pin_out + (size_t)n_h * 8, sizeof best);
  if (best_index) *best_index = best[0];
  if (best_score) std::memcpy(best_score, &best[1], 4);
  m->last_ms[1] = (float)(t1 - t0);
  m->last_ms[2] = (float)(now_ms() - t0);
  return PGP_OK;
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

}  // extern "C"
