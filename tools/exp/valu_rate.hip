// Issue cost of the vector instructions a GEMM epilogue is made of, on one CU at the GEMM kernels' occupancy (8 waves = 2 per SIMD) and at 1 wave per SIMD:
// cycles per instruction and wave (s_memtime around N repetitions of a 16-instruction body, the slowest wave of the workgroup).  Round 5: is GELU's cost in the
// lin1 epilogue its transcendentals, its packed-fp32 chains (one wait state between dependent v_pk_* instructions) or neither?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/valu_rate.hip -o tools/exp/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 0.1f, a2 = a0 + 0.2f, a3 = a0 + 0.3f, a4 = a0 + 0.4f, a5 = a0 + 0.5f, a6 = a0 + 0.6f, a7 = a0 + 0.7f;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const f2 c = {0.999f, 0.999f}, d = {1e-3f, 1e-3f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {        // 16 independent-ish v_fma_f32 (8 chains)
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(a0) : "v"(a1), "v"(a2));)
    } else if constexpr (KIND == 1) { // v_exp_f32 back to back, 8 registers in turn
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 2) { // v_pk_fma_f32, four independent chains in turn (no wait state needed between different chains)
      asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(d));
    } else if constexpr (KIND == 3) { // ONE dependent v_pk_fma_f32 chain with the s_nop 0 the compiler puts between them: 8 x (pk_fma + nop)
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n"
                   "v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n"
                   : "+v"(p0) : "v"(c), "v"(d));
    } else if constexpr (KIND == 4) { // v_exp_f32 whose result the next instruction consumes (exp -> fma -> exp -> fma ...): 8 + 8
      asm volatile("v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n v_exp_f32 %0, %0\n s_nop 0\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(a0) : "v"(a1), "v"(a2));
    } else if constexpr (KIND == 5) { // v_cvt_pk_f16_f32 / v_cvt_pk_bf8_f32-like conversions: 16 v_cvt_pk_f16_f32 (v_cvt_pkrtz)
      REP16(asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2\n" : "=v"(a0) : "v"(a1), "v"(a2));)
    } else if constexpr (KIND == 6) { // v_permlane16_swap
      REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n" : "+v"(a0), "+v"(a1));)
    } else if constexpr (KIND == 7) { // v_rcp_f32 back to back on 8 registers
      asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                   "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if constexpr (KIND == 8) { // v_pk_mul_f32 four chains
      asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_body) {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
  const int iters = 2000;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, cyc, 8 * (threads >> 6), hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < (threads >> 6); ++i) mx = h[i] > mx ? h[i] : mx;
    printf("%-48s %d wave(s)/SIMD: %7.2f s_memtime cycles per instruction and wave (x waves per SIMD = SIMD cycles)\n", name, threads >> 8, (double)mx / ((double)iters * per_body));
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  int wall = 0; hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("shader clock %d kHz, wall clock (s_memtime) %d kHz: 1 tick = %.1f shader cycles at the nominal clock\n", clk, wall, wall ? (double)clk / wall : 0.0);
  run<0>("v_fma_f32 (dependent chain)", 16);
  run<1>("v_exp_f32 (8 registers in turn)", 16);
  run<7>("v_rcp_f32 (8 registers in turn)", 16);
  run<2>("v_pk_fma_f32 (4 chains in turn)", 16);
  run<8>("v_pk_mul_f32 (4 chains in turn)", 16);
  run<3>("v_pk_fma_f32 + s_nop 0 (one chain)", 8);
  run<4>("v_exp_f32 -> s_nop 0 -> v_fma_f32 (one chain)", 8);
  run<5>("v_cvt_pkrtz_f16_f32", 16);
  run<6>("v_permlane16_swap_b32", 16);
  return 0;
}
