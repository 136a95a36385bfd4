// shim/frame_pool.h -- the kept worker threads of getProbableTransformsSuper4PCSFrame, free of Eigen and HIP so that the
// same code is built under ThreadSanitizer / AddressSanitizer without a GPU (shim/test_concurrency.cc).
//
// The node matches the objects of a frame one after the other (SceneCfg.cpp:379-402 -> ObjectPoseCandidateSet.cpp:53-68); a
// call is a chain of short device steps with the host in between, so one object leaves most of the GPU and most of the call's
// wall-clock unused.  Every job of a frame runs on a thread the process keeps (job k on worker k % kWorkers); one frame at a
// time (use_mu).
#pragma once

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

namespace shimstate {

class FramePool {
 public:
  static const int kWorkers = 8;
  std::mutex use_mu;   // one frame at a time
  // on_thread_start runs once on every worker thread (the drop-in marks them as drawing from a generator of their own)
  static FramePool* make(std::function<void()> on_thread_start) {
    FramePool* p = new FramePool;   // never destroyed: the workers sleep on their condition variables until the process ends
    p->on_thread_start_ = std::move(on_thread_start);
    try {
      for (int k = 0; k < kWorkers; ++k) std::thread([p, k] { p->loop(k); }).detach();
    } catch (...) {
      return nullptr;   // (workers already started sleep for good; the frame is then matched job by job on the caller's thread)
    }
    return p;
  }
  void start(int k, std::function<void()> fn) {
    Worker& w = workers_[k];
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.job = std::move(fn);
      w.has_job = true;
      w.done = false;
    }
    w.cv.notify_all();
  }
  void wait(int k) {
    Worker& w = workers_[k];
    std::unique_lock<std::mutex> lk(w.mu);
    w.cv.wait(lk, [&] { return w.done; });
  }

 private:
  struct Worker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true;
  };
  Worker workers_[kWorkers];
  std::function<void()> on_thread_start_;
  void loop(int k) {
    Worker& w = workers_[k];
    if (on_thread_start_) on_thread_start_();
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.has_job; });
        fn.swap(w.job);
        w.has_job = false;
      }
      fn();
      {
        std::lock_guard<std::mutex> lk(w.mu);
        w.done = true;
      }
      w.cv.notify_all();
    }
  }
};

}  // namespace shimstate
