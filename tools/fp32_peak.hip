// tools/fp32_peak.hip -- which FP32 vector peak is real on gfx950?  (task 6)
// /opt/skills/guides/MI355X_MICROARCH.md lists "v_fma_f32 (wave64) 2 cyc (SIMD-32)" AND a vector FP32 peak of
// 157.3 TFLOP/s = 64 FLOP/clk/SIMD; tools/valu_rates.hip measured 4.0 cycles per wave64 v_fma_f32 (and 4.4 per
// v_pk_fma_f32) assuming the nominal 2.4 GHz.  Both cannot be the unit of a roofline.  This program runs
// independent v_fma_f32 and v_pk_fma_f32 streams (inline asm, 8 accumulators per lane) at 8 waves per SIMD on
// every CU for ~0.4 s each, reads the clock the chip actually holds (s_memtime ticks per 100 MHz s_memrealtime
// tick, median over workgroups) and prints FLOP/s, instructions/s and cycles per instruction AT THAT CLOCK.
//   hipcc -O3 --offload-arch=gfx950 tools/fp32_peak.hip -o /tmp/fp32_peak && /tmp/fp32_peak
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int kPerTrip = 32;

template <int KIND>
__global__ __launch_bounds__(256) void stream_kernel(float* out, unsigned long long* stamps, float a, float b, int iters) {
  float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b}, p4 = {a, b}, p5 = {b, a}, p6 = {a, a}, p7 = {b, b};
  f2 src = {b, a};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)
    if (KIND == 0) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "v"(src));
      REP32(OP)
#undef OP
    } else {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
  const float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static void measure(const char* name, int flop_per_lane_instr, int cus, float* out, unsigned long long* d_st, bool last) {
  const int blocks = cus * 8, iters = 1 << 17;   // 8 waves per SIMD; 4.2 M instructions per wave
  std::vector<unsigned long long> st(2 * (size_t)blocks);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(stream_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, d_st, 1.0f, 1e-9f, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  const int reps = 4;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(stream_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, d_st, 1.0f, 1e-9f, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int b = 0; b < blocks; ++b)
    if (st[2 * b + 1] > 0) clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100e6);
  std::sort(clk.begin(), clk.end());
  const double clock_hz = clk.empty() ? 0.0 : clk[clk.size() / 2];
  const double wave_instr = (double)reps * blocks * 4.0 * iters * kPerTrip;   // 4 waves per block
  const double t = ms * 1e-3;
  const double instr_per_s_per_simd = wave_instr / t / (cus * 4.0);
  printf("\"%s\": {\"seconds\": %.3f, \"clock_GHz_in_kernel\": %.3f, \"wave_instr_per_s\": %.4g, \"TFLOP_per_s\": %.2f, "
         "\"cycles_per_wave_instr_per_simd\": %.3f, \"cycles_at_nominal_2.4GHz\": %.3f}%s\n",
         name, t, clock_hz * 1e-9, wave_instr / t, wave_instr * 64.0 * flop_per_lane_instr / t * 1e-12,
         clock_hz / instr_per_s_per_simd, 2.4e9 / instr_per_s_per_simd, last ? "" : ",");
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  float* out;
  unsigned long long* d_st;
  if (hipMalloc(&out, 4096) != hipSuccess || hipMalloc(&d_st, (size_t)p.multiProcessorCount * 8 * 16) != hipSuccess) return 1;
  printf("{\"device\": \"%s\", \"cus\": %d, \"nominal_clock_mhz\": %d, \"spec_fp32_vector_TFLOPs\": 157.3,\n", p.name,
         p.multiProcessorCount, p.clockRate / 1000);
  measure<0>("v_fma_f32", 2, p.multiProcessorCount, out, d_st, false);
  measure<1>("v_pk_fma_f32", 4, p.multiProcessorCount, out, d_st, false);
  measure<2>("v_add_f32", 1, p.multiProcessorCount, out, d_st, true);
  printf("}\n");
  return 0;
}
