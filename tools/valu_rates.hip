// tools/valu_rates.hip -- issue cost of individual gfx950 VALU instructions, in cycles per wave64
// instruction per SIMD, measured with inline asm (so the compiler can neither pack nor fuse anything).
// tools/peaks.hip's "v_add_f32" figure was taken from C code that hipcc turned into v_pk_add_f32 (two
// adds per instruction): this file is the correction.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o gpurun_out/valu_rates && ./gpurun_out/valu_rates
// Prints one JSON object: {"<instr>": {"cycles_per_instr_8waves": c8, "cycles_per_instr_1wave": c1}, ...}
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 4096;
constexpr int kPerTrip = 32;   // instructions per loop trip (independent destinations)

// 8 independent 2-register destinations v[d:d+1]; sources are loop-invariant.
#define REP8(OP)                                                                                   \
  OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, float a, float b, int c) {
  float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a, b}, p1 = {b, a}, p2 = {a, a}, p3 = {b, b}, p4 = {a, b}, p5 = {b, a}, p6 = {a, a}, p7 = {b, b};
  f2 src = {b, a};
  unsigned long long sink = 0, sel = (unsigned long long)c * 0x5555555555ull;
  if (KIND == 24) asm volatile("s_mov_b64 vcc, %0" : : "s"(sel) : "vcc");
  for (int it = 0; it < kIters; ++it) {
    if (KIND == 0) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 1) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 2) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 3) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##i) : "v"(src));
      REP32(OP)
#undef OP
    } else if (KIND == 4) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(src));
      REP32(OP)
#undef OP
    } else if (KIND == 5) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p##i) : "v"(src));
      REP32(OP)
#undef OP
    } else if (KIND == 6) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 7) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 8) {
#define OP(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r##i));
      REP32(OP)
#undef OP
    } else if (KIND == 9) {
#define OP(i) asm volatile("v_med3_i32 %0, %0, 0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 10) {
#define OP(i) asm volatile("v_bfe_u32 %0, %0, 2, 8" : "+v"(r##i));
      REP32(OP)
#undef OP
    } else if (KIND == 11) {
#define OP(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 12) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(c) : );
      REP32(OP)
#undef OP
    } else if (KIND == 13) {
#define OP(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r##i));
      REP32(OP)
#undef OP
    } else if (KIND == 14) {
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(p##i) : "v"(src));
      REP32(OP)
#undef OP
    } else if (KIND == 15) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r##i), "v"(b) : "vcc");
      REP32(OP)
#undef OP
    } else if (KIND == 16) {
#define OP(i) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 17) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 18) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 19) {
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(c) : "v"(r##i));
      REP32(OP)
#undef OP
    } else if (KIND == 20) {
      // packed multiply with a scalar pair and broadcast of its low half (op_sel_hi:[1,0]): the
      // shape of "matrix element (SGPR) x two model points (VGPR pair)"
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p##i) : "s"(src));
      REP32(OP)
#undef OP
    } else if (KIND == 21) {
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 22) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 23) {
#define OP(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 24) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r##i) : "v"(c) : );
      REP32(OP)
#undef OP
    }    else if (KIND == 25) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r##i) : "v"(c), "s"(sel));
      REP32(OP)
#undef OP
    }    else if (KIND == 26) {
#define OP(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 27) {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 28) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r##i));
      REP32(OP)
#undef OP
    }    else if (KIND == 29) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r##i));
      REP32(OP)
#undef OP
    }    else if (KIND == 30) {
#define OP(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 31) {
#define OP(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 32) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 33) {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 34) {
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 35) {
#define OP(i) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 36) {
#define OP(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sel) : "v"(r##i), "v"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 37) {
#define OP(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 38) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 39) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r##i) : "s"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 40) {
#define OP(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r##i));
      REP32(OP)
#undef OP
    }    else if (KIND == 41) {
#define OP(i) asm volatile("v_floor_f32 %0, %0" : "+v"(r##i));
      REP32(OP)
#undef OP
    }    else if (KIND == 42) {
#define OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 43) {
#define OP(i) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(r##i) : "s"(b));
      REP32(OP)
#undef OP
    }    else if (KIND == 44) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1\n v_mad_u32_u24 %0, %0, %2, %0" : "+v"(r##i) : "v"(b), "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 45) {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 46) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    }    else if (KIND == 47) {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(r##i) : "v"(c));
      REP32(OP)
#undef OP
    } else if (KIND == 48) {
#define OP(i) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
      REP32(OP)
#undef OP
    } else if (KIND == 49) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_add_u32 s20, s20, 1" : "+v"(r##i) : "v"(b) : "s20", "scc");
      REP32(OP)
#undef OP
    } else if (KIND == 50) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1\n s_add_u32 s20, s20, 1" : "+v"(r##i) : "v"(b) : "s20", "scc");
      REP32(OP)
#undef OP
    } else if (KIND == 51) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1" : "+v"(r##i) : "v"(b) : "s20", "scc");
      REP32(OP)
#undef OP
    } else if (KIND == 52) {
#define OP(i) asm volatile("s_nop 0");
      REP32(OP)
#undef OP
    } else if (KIND == 53) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_nop 0" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    } else if (KIND == 54) {
#define OP(i) asm volatile("s_mov_b64 s[20:21], s[22:23]" ::: "s20", "s21");
      REP32(OP)
#undef OP
    } else if (KIND == 55) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %1\n s_waitcnt lgkmcnt(0)" : "+v"(r##i) : "v"(b));
      REP32(OP)
#undef OP
    }
  }
  float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)c +
            (float)sink + (float)sel;
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static const char* kNames[] = {"v_add_f32",      "v_mul_f32",     "v_fma_f32",      "v_pk_add_f32",   "v_pk_mul_f32",
                               "v_pk_fma_f32",   "v_add_u32",     "v_mad_u32_u24",  "v_cvt_i32_f32",  "v_med3_i32",
                               "v_bfe_u32",      "v_bcnt_u32_b32", "v_cndmask_b32", "v_add_u32_dpp",  "v_lshl_add_u64",
                               "v_cmp_lt_f32",   "v_lshl_or_b32", "v_and_b32",      "v_mul_lo_u32",   "v_readlane_b32",
                               "v_pk_mul_f32_sgpr_bcast", "v_med3_f32", "v_mov_b32", "v_mbcnt_lo_u32_b32",
                               "v_cndmask_b32_vcc_set", "v_cndmask_b32_e64_sgpr", "v_add_f32_e64", "v_sub_f32", "v_lshlrev_b32", "v_lshrrev_b32", "v_or_b32", "v_min_u32", "v_max_f32", "v_fmac_f32", "v_add3_u32", "v_or3_b32", "v_cmp_lt_f32_e64_sgpr", "v_bfi_b32", "v_mul_u32_u24", "v_mul_f32_sgpr_src", "v_cvt_u32_f32", "v_floor_f32", "v_xor_b32", "v_subrev_f32_sgpr", "v_add_f32_mixed_chain", "v_perm_b32", "v_and_or_b32", "v_lshl_add_u32",
                               "s_add_u32_alone", "v_fma_f32_plus_s_add_u32", "v_add_f32_plus_s_add_u32", "v_fma_f32_plus_2_s_add_u32", "s_nop_0", "v_fma_f32_plus_s_nop", "s_mov_b64", "v_fma_f32_plus_s_waitcnt"};
constexpr int kKinds = 56;

template <int KIND>
static float run(int blocks, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-9f, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1e-9f, 3);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

template <int KIND>
static void all(int cus, double clock_hz, float* out) {
  // 8 waves per SIMD: 8 blocks of 4 waves per CU; 1 wave per SIMD: 1 block per CU
  const float ms8 = run<KIND>(cus * 8, out);
  const float ms1 = run<KIND>(cus, out);
  const double instr_per_simd8 = 8.0 * kIters * kPerTrip, instr_per_simd1 = 1.0 * kIters * kPerTrip;
  printf("%s\"%s\": {\"cycles_per_instr_8waves\": %.2f, \"cycles_per_instr_1wave\": %.2f}", KIND ? ", " : "",
         kNames[KIND], ms8 * 1e-3 * clock_hz / instr_per_simd8, ms1 * 1e-3 * clock_hz / instr_per_simd1);
  fflush(stdout);
  if constexpr (KIND + 1 < kKinds) all<KIND + 1>(cus, clock_hz, out);
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 1;
  float* out;
  if (hipMalloc(&out, 4096) != hipSuccess) return 1;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"rates\": {", p.name, p.multiProcessorCount,
         p.clockRate / 1000);
  all<0>(p.multiProcessorCount, (double)p.clockRate * 1e3, out);
  printf("}}\n");
  return 0;
}
