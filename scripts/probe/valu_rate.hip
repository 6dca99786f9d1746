// valu_rate -- GPU-box microbenchmark (diagnostics, not product): SIMD cycles per wave64 instruction on gfx950 for
// the instructions K1's epilogue is made of, (a) as independent streams, (b) as one dependent chain, at 1 / 2 / 3
// waves per SIMD, alone and with one v_mfma_f32_32x32x16_bf16 per 14 instructions (K1's mix).
// Build: make -C scripts/probe valu_rate.   Prints one JSON line per case.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X X X X X X X X
#define REP56(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

// CASE: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_alignbit_b32, 3 v_min3_f32 |abs|, 4 v_bfi_b32, 5 v_perm_b32,
// 6 v_mov_b32_dpp quad_perm, 7 v_mov_b32_dpp row_ror:8, 8 v_permlane16_swap, 9 v_cmp + v_cndmask, 10 v_lshl_or_b32,
// 11 v_and_b32 (VOP2), 12 v_bcnt_u32_b32, 13 v_fmac_f32 (VOP2), 14 v_add_u32 (VOP2), 15 v_fma_f32 with |abs| src
template <int CASE, bool DEP, bool MFMA>
__global__ __launch_bounds__(256) void rate_kernel(unsigned int* out, int iters, unsigned int seed) {
  unsigned int a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u,
               a7 = a0 * 19u;
  float f0 = (float)a0, f1 = f0 * 1.1f, f2 = f0 * 1.2f, f3 = f0 * 1.3f, f4 = f0 * 1.4f, f5 = f0 * 1.5f, f6 = f0 * 1.6f,
        f7 = f0 * 1.7f;
  const unsigned int k = seed | 1u;
  const float kf = 1.0000001f;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 ma, mb;
  for (int i = 0; i < 8; ++i) {
    ma[i] = (__bf16)(float)(threadIdx.x + i);
    mb[i] = (__bf16)1.0f;
  }
#define OP_U(dst, src)                                                                                              \
  if (CASE == 2) asm volatile("v_alignbit_b32 %0, %1, %2, 31" : "+v"(dst) : "v"(dst), "v"(src));                   \
  if (CASE == 4) asm volatile("v_bfi_b32 %0, %2, %1, %0" : "+v"(dst) : "v"(src), "v"(k));                         \
  if (CASE == 5) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(dst) : "v"(src), "v"(k));                        \
  if (CASE == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src)); \
  if (CASE == 7) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src)); \
  if (CASE == 8) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(dst), "+v"(src));                             \
  if (CASE == 10) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(dst) : "v"(src));                             \
  if (CASE == 11) asm volatile("v_and_b32 %0, %1, %0" : "+v"(dst) : "v"(src));                                    \
  if (CASE == 12) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(dst) : "v"(src));                               \
  if (CASE == 14) asm volatile("v_add_u32 %0, %1, %0" : "+v"(dst) : "v"(src));
#define OP_F(dst, src)                                                                                              \
  if (CASE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(dst) : "v"(src), "v"(kf));                        \
  if (CASE == 3) asm volatile("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(dst) : "v"(src), "v"(kf));                   \
  if (CASE == 9) asm volatile("v_cmp_ngt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(dst) : "v"(src) : "vcc"); \
  if (CASE == 13) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(dst) : "v"(src), "v"(kf));                          \
  if (CASE == 15) asm volatile("v_fma_f32 %0, %1, %2, |%0|" : "+v"(dst) : "v"(src), "v"(kf));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
  const f32x2 kp = {kf, kf};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // 8 x 14 instructions (+ 8 MFMAs)
      if (MFMA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, acc, 0, 0, 0);
      if (CASE == 1) {
        if (DEP) {
          for (int q = 0; q < 14; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p0) : "v"(kp));
        } else {
          for (int q = 0; q < 3; ++q) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p0) : "v"(kp));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p1) : "v"(kp));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p2) : "v"(kp));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p3) : "v"(kp));
          }
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p0) : "v"(kp));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p1) : "v"(kp));
        }
      } else if (DEP) {
        for (int q = 0; q < 14; ++q) {
          OP_U(a0, a1)
          OP_F(f0, f1)
        }
      } else {
        OP_U(a0, a1) OP_U(a2, a3) OP_U(a4, a5) OP_U(a6, a7) OP_U(a1, a2) OP_U(a3, a4) OP_U(a5, a6)
        OP_U(a7, a0) OP_U(a0, a3) OP_U(a2, a5) OP_U(a4, a7) OP_U(a6, a1) OP_U(a1, a4) OP_U(a3, a6)
        OP_F(f0, f1) OP_F(f2, f3) OP_F(f4, f5) OP_F(f6, f7) OP_F(f1, f2) OP_F(f3, f4) OP_F(f5, f6)
        OP_F(f7, f0) OP_F(f0, f3) OP_F(f2, f5) OP_F(f4, f7) OP_F(f6, f1) OP_F(f1, f4) OP_F(f3, f6)
      }
    }
  }
  unsigned int s = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  float fs = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  for (int i = 0; i < 16; ++i) fs += acc[i];
  if (s == 0x12345u && fs == 1.234f) out[0] = s;  // keep everything alive
}

template <int CASE, bool DEP, bool MFMA>
static void run(const char* name, int waves_per_simd, unsigned int* d_out, double ghz, int cus) {
  const int iters = 2000;
  // blocks of 256 threads = one wave per SIMD of a CU; `waves_per_simd` blocks per CU
  dim3 grid(cus * waves_per_simd), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<CASE, DEP, MFMA>), grid, block, 0, 0, d_out, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<CASE, DEP, MFMA>), grid, block, 0, 0, d_out, iters, 1u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_wave = (double)iters * 8 * 14 * (CASE == 9 ? 2 : 1);
  const double cyc = ms * 1e-3 * ghz * 1e9;  // per SIMD (all SIMDs run the same)
  printf("{\"probe\":\"valu_rate\",\"inst\":\"%s\",\"dependent\":%d,\"with_mfma\":%d,\"waves_per_simd\":%d,\"ms\":%.4f,"
         "\"cycles_per_inst_per_simd\":%.3f,\"cycles_per_group_of_14\":%.2f}\n",
         name, DEP ? 1 : 0, MFMA ? 1 : 0, waves_per_simd, ms, cyc / (insts_per_wave * waves_per_simd),
         cyc / ((double)iters * 8 * waves_per_simd));
  fflush(stdout);
}

#define RUN_ALL(CASE, NAME)                                  \
  for (int w : {1, 3}) {                                     \
    run<CASE, false, false>(NAME, w, d_out, ghz, cus);       \
    run<CASE, true, false>(NAME, w, d_out, ghz, cus);        \
    run<CASE, false, true>(NAME, w, d_out, ghz, cus);        \
  }

int main(int argc, char** argv) {
  const double ghz = argc > 1 ? atof(argv[1]) : 2.4;  // nominal; the JSON keeps ms so any clock can be applied
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  unsigned int* d_out;
  hipMalloc(&d_out, 64);
  printf("{\"probe\":\"valu_rate\",\"cus\":%d,\"clock_mhz\":%d,\"assumed_ghz\":%.2f}\n", cus, prop.clockRate / 1000, ghz);
  RUN_ALL(0, "v_fma_f32")
  RUN_ALL(15, "v_fma_f32 |abs|")
  RUN_ALL(13, "v_fmac_f32")
  RUN_ALL(1, "v_pk_fma_f32")
  RUN_ALL(2, "v_alignbit_b32")
  RUN_ALL(3, "v_min3_f32 |abs|")
  RUN_ALL(4, "v_bfi_b32")
  RUN_ALL(5, "v_perm_b32")
  RUN_ALL(6, "v_mov_b32_dpp quad_perm")
  RUN_ALL(7, "v_mov_b32_dpp row_ror")
  RUN_ALL(8, "v_permlane16_swap_b32")
  RUN_ALL(9, "v_cmp_ngt_f32 + v_cndmask_b32")
  RUN_ALL(10, "v_lshl_or_b32")
  RUN_ALL(11, "v_and_b32")
  RUN_ALL(12, "v_bcnt_u32_b32")
  RUN_ALL(14, "v_add_u32")
  return 0;
}
