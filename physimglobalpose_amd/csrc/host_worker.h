// csrc/host_worker.h -- the host-side hand-offs of the device group (multi_gpu.hip), free of HIP so that the same code is
// built under ThreadSanitizer / AddressSanitizer on a machine without a GPU (shim/test_concurrency.cc, `make -C shim tsan`).
//
//   Worker          one kept host thread per device of a group: the calling thread posts a job and waits for it; both
//                   hand-offs spin on an atomic flag for a short while before they sleep on the condition variable (a scoring
//                   call is ~0.1 ms of GPU work, a futex wake-up 20-40 us each way)
//   PerDeviceTable  one lazily created object per device id shared by every context of the process (the side stream of the
//                   index builds, grid_index.hip): created once under a lock, never destroyed; ids beyond the table get none
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>

namespace pgp {

struct Worker {
  int device = 0;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  std::atomic<bool> has_job{false}, done{true};
  bool stop = false;
  int rc = 0;
  char err[512] = "";
  std::function<void()> on_start;            // runs first on the worker's thread (hipSetDevice)
  std::function<const char*()> last_error;   // the thread's error text after a job that returned non-zero (pgp_last_error)
  static constexpr double kSpinWorkerMs = 0.2, kSpinCallerMs = 2.0;

  template <class Pred>
  static bool spin(double ms, Pred p) {
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::duration<double, std::milli>(ms);
    for (;;) {
      for (int i = 0; i < 64; ++i) {
        if (p()) return true;
        __builtin_ia32_pause();
      }
      if (std::chrono::steady_clock::now() >= t_end) return p();
    }
  }
  void loop() {
    if (on_start) on_start();
    for (;;) {
      std::function<int()> j;
      spin(kSpinWorkerMs, [&] { return has_job.load(std::memory_order_acquire); });
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return has_job.load(std::memory_order_acquire) || stop; });
        if (stop) return;
        j = std::move(job);
        has_job.store(false, std::memory_order_relaxed);
      }
      int r = j();
      {
        std::lock_guard<std::mutex> lk(mu);
        rc = r;
        if (r != 0 && last_error) {
          std::strncpy(err, last_error(), sizeof err - 1);
          err[sizeof err - 1] = 0;
        }
        done.store(true, std::memory_order_release);
      }
      cv.notify_all();
    }
  }
  // one caller at a time (calls on one group must not overlap): post, then wait
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      done.store(false, std::memory_order_relaxed);
      has_job.store(true, std::memory_order_release);
    }
    cv.notify_all();
  }
  int wait() {
    spin(kSpinCallerMs, [&] { return done.load(std::memory_order_acquire); });
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done.load(std::memory_order_acquire); });
    return rc;
  }
  void shut_down() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

template <class T, int N = 64>
struct PerDeviceTable {
  std::mutex mu;
  T slot[N] = {};
  bool made[N] = {};
  // the device's shared object, created by `create` (returns false on failure) the first time; false when the id is outside
  // the table -- the caller then makes an object of its own (a device >= N must not borrow device 0's: ADVICE r5)
  template <class Create>
  bool get(int device, Create create, T* out) {
    if (device < 0 || device >= N) return false;
    std::lock_guard<std::mutex> lk(mu);
    if (!made[device]) {
      if (!create(&slot[device])) return false;
      made[device] = true;
    }
    *out = slot[device];
    return true;
  }
};

}  // namespace pgp
