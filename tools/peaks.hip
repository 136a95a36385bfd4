// tools/peaks.hip -- measured ceilings of the two units that bind the scoring kernel:
// VALU issue (wave-instructions/s at 8 waves per SIMD) and the vector L1 (cache-line accesses/s,
// L1-resident data).  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/peaks.hip -o gpurun_out/peaks && ./gpurun_out/peaks
// Prints one JSON object.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                               \
    }                                                                         \
  } while (0)

constexpr int kIters = 2048;

// 64 independent VALU ops per trip on 16 accumulators: v_add_f32 / v_mul_f32 (no contraction).
template <int KIND>
__global__ __launch_bounds__(256) void valu_kernel(float* out, float a, float b) {
  float r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = a + (float)(threadIdx.x + i);
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) r[i] = __fadd_rn(r[i], b);
        else if (KIND == 1) r[i] = __fmul_rn(r[i], b);
        else r[i] = __int_as_float((__float_as_int(r[i]) + __float_as_int(b)) ^ (int)threadIdx.x);   // 2 integer ops
      }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// Each load instruction of a wave touches `LINES` distinct 64-byte lines of a small (L1-resident)
// per-block buffer; 8 independent loads per trip.
template <int STRIDE_DW>
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ buf, float* out, int words_per_block) {
  const float* base = buf + (size_t)(blockIdx.x % 1024) * words_per_block;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = (lane * STRIDE_DW + wave * 4096) % words_per_block;
  float s = 0.f;
  for (int it = 0; it < kIters / 4; ++it) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base[(idx + k * 1024) % words_per_block];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    idx = (idx + 16) % words_per_block;
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// The same with 8- and 16-byte loads (the scoring kernel's word / run-descriptor loads are 8 B per
// lane, its candidate gathers 16 B per lane): VEC floats per lane, lanes STRIDE_VEC vectors apart.
template <int VEC, int STRIDE_VEC>
__global__ __launch_bounds__(256) void l1v_kernel(const float* __restrict__ buf, float* out, int words_per_block) {
  typedef float vec_t __attribute__((ext_vector_type(VEC)));
  const vec_t* base = reinterpret_cast<const vec_t*>(buf + (size_t)(blockIdx.x % 1024) * words_per_block);
  const int nv = words_per_block / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int idx = (lane * STRIDE_VEC + wave * (nv / 4)) % nv;
  float s = 0.f;
  for (int it = 0; it < kIters / 4; ++it) {
    vec_t v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = base[(idx + k * (nv / 8)) % nv];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k][0] + v[k][VEC - 1];
    idx = (idx + 64 / (4 * VEC)) % nv;
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <class F>
static float time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  float* out;
  CK(hipMalloc(&out, 4096));
  const int blocks = cus * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", p.name, cus, p.clockRate / 1000);
  const char* names[3] = {"v_add_f32", "v_mul_f32", "int_add_xor"};
  for (int kind = 0; kind < 3; ++kind) {
    float ms;
    if (kind == 0) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-9f); }, 5);
    else if (kind == 1) ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1.0000001f); }, 5);
    else ms = time_ms([&] { hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-9f); }, 5);
    const double instr = (double)blocks * 4 * kIters * 64 * (kind == 2 ? 2 : 1);
    printf(", \"%s_Gwaveinstr_per_s\": %.1f", names[kind], instr / (ms * 1e-3) / 1e9);
  }
  // vector L1: 16 KB per block slot (L1 = 32 KB per CU), 1024 slots = 16 MB buffer
  const int wpb = 4096;
  float* buf;
  CK(hipMalloc(&buf, (size_t)1024 * wpb * 4));
  CK(hipMemset(buf, 0, (size_t)1024 * wpb * 4));
  {
    // stride 16 dwords = 64 B: every lane its own line -> 64 lines per instruction
    float ms = time_ms([&] { hipLaunchKernelGGL(l1_kernel<16>, dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    const double instr = (double)blocks * 4 * (kIters / 4) * 8;
    printf(", \"l1_scattered_Ginstr_per_s\": %.2f, \"l1_scattered_Glines_per_s\": %.1f", instr / (ms * 1e-3) / 1e9,
           instr * 64 / (ms * 1e-3) / 1e9);
    // stride 1 dword: a wave reads 256 contiguous bytes = 4 lines per instruction
    ms = time_ms([&] { hipLaunchKernelGGL(l1_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    printf(", \"l1_coalesced_Ginstr_per_s\": %.2f, \"l1_coalesced_Glines_per_s\": %.1f", instr / (ms * 1e-3) / 1e9,
           instr * 4 / (ms * 1e-3) / 1e9);
    // stride 4 dwords = 16 B: 16 lines per instruction (the shape of a 64-candidate float4 gather)
    ms = time_ms([&] { hipLaunchKernelGGL(l1_kernel<4>, dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    printf(", \"l1_16B_stride_Ginstr_per_s\": %.2f, \"l1_16B_stride_Glines_per_s\": %.1f", instr / (ms * 1e-3) / 1e9,
           instr * 16 / (ms * 1e-3) / 1e9);
    struct V { const char* name; float ms; };
    const double ins = instr;
    float m2s = time_ms([&] { hipLaunchKernelGGL((l1v_kernel<2, 8>), dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    float m4s = time_ms([&] { hipLaunchKernelGGL((l1v_kernel<4, 4>), dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    float m4c = time_ms([&] { hipLaunchKernelGGL((l1v_kernel<4, 1>), dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    float m2c = time_ms([&] { hipLaunchKernelGGL((l1v_kernel<2, 1>), dim3(blocks), dim3(256), 0, 0, buf, out, wpb); }, 5);
    printf(", \"l1_8B_one_line_per_lane_Ginstr_per_s\": %.2f, \"l1_16B_one_line_per_lane_Ginstr_per_s\": %.2f, "
           "\"l1_16B_contiguous_Ginstr_per_s\": %.2f, \"l1_8B_contiguous_Ginstr_per_s\": %.2f",
           ins / (m2s * 1e-3) / 1e9, ins / (m4s * 1e-3) / 1e9, ins / (m4c * 1e-3) / 1e9, ins / (m2c * 1e-3) / 1e9);
  }
  printf("}\n");
  return 0;
}
