// tools/copy_cost.hip -- host time of small copies from / to pageable and pinned memory (what a host-pointer C-ABI call pays
// per staging copy): hipMemcpyAsync + hipStreamSynchronize, median of 2000.
// build: hipcc -O2 --offload-arch=gfx950 tools/copy_cost.hip -o tools/ab/copy_cost
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void nop(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  void* d;
  hipMalloc(&d, 1 << 22);
  void* pin;
  hipHostMalloc(&pin, 1 << 22, hipHostMallocDefault);
  std::vector<unsigned char> page(1 << 22, 1);
  const size_t sizes[] = {64, 4096, 65536, 524288};
  for (size_t n : sizes) {
    for (int dir = 0; dir < 2; ++dir) {
      for (int pinned = 0; pinned < 3; ++pinned) {   // 2: memcpy through the pinned buffer (what a bounce costs in all)
        std::vector<double> t;
        for (int it = 0; it < 2000; ++it) {
          hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, st, (int*)nullptr);   // the stream is busy, as in a call
          const double t0 = now_us();
          void* h = pinned ? pin : (void*)page.data();
          if (pinned == 2 && dir == 0) std::memcpy(pin, page.data(), n);
          if (dir == 0) hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, st);
          else hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, st);
          const double t1 = now_us();
          hipStreamSynchronize(st);
          if (pinned == 2 && dir == 1) std::memcpy(page.data(), pin, n);
          const double t2 = now_us();
          t.push_back(t2 - t0);
          (void)t1;
        }
        std::sort(t.begin(), t.end());
        printf("%7zu B %s %-22s median %6.1f us  p90 %6.1f\n", n, dir ? "D2H" : "H2D",
               pinned == 0 ? "pageable" : (pinned == 1 ? "pinned" : "pageable via pinned"), t[t.size() / 2], t[t.size() * 9 / 10]);
      }
    }
  }
  // issue cost alone (no sync): how long the host is held by the call
  for (int pinned = 0; pinned < 2; ++pinned) {
    std::vector<double> t;
    for (int it = 0; it < 2000; ++it) {
      const double t0 = now_us();
      hipMemcpyAsync(d, pinned ? pin : (void*)page.data(), 65536, hipMemcpyHostToDevice, st);
      t.push_back(now_us() - t0);
      hipStreamSynchronize(st);
    }
    std::sort(t.begin(), t.end());
    printf("issue of a 64 KB H2D, %s: median %.1f us\n", pinned ? "pinned" : "pageable", t[t.size() / 2]);
  }
  return 0;
}
