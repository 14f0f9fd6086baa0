// Does the L2 -> LDS operand stream of the GEMM's ping-pong loop cost BYTES or INSTRUCTIONS?  The loop of pp_conflict.hip (mode 9: the 8
// waves' LDS-DMA alone, 6 instructions of 64 lanes x 16 B per wave and iteration = 48 KiB per CU; mode 8: beside 32 MFMAs of the partner
// group) with only `pieces` of every row's 8 sixteen-byte pieces requested (lanes with (lane & 7) >= pieces are masked off: same
// instruction count, same LDS row stride, fewer bytes).  h8 operands need 6 of the 8 pieces of a line if q(hi) is derived from the hi
// fragment in registers instead of being fetched.   hipcc --offload-arch=gfx950 -O3 tools/exp/dma_lanes.hip -o /tmp/dma_lanes && /tmp/dma_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, int pieces, const unsigned char* src) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  for (int i = threadIdx.x; i < 40960; i += 512) reinterpret_cast<float*>(smem)[i] = (float)(i & 255) * 0.001f;
  __syncthreads();
  const int l15 = lane & 15, g = lane >> 4;
  const unsigned off = (wave & 3) * 8192 + l15 * 128 + ((g ^ ((l15 >> 1) & 7)) * 16);
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fr[16];
  for (int i = 0; i < 16; ++i) fr[i] = *reinterpret_cast<const bf16x8*>(smem + off + (i & 3) * 2048);
  const unsigned char* gsrc = src + (size_t)(blockIdx.x & 63) * (2u << 20) + (size_t)wave * 6 * 1024 + lane * 16;
  const bool on = (lane & 7) < pieces;
  if (grp) __builtin_amdgcn_s_barrier();
  long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* gp = gsrc + (size_t)(it & 31) * 49152;
    unsigned char* l = smem + (it % 3) * 49152 + wave * 6 * 1024;
    if (on) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + i * 1024), (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (mode == 8) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i & 7], fr[8 + (i & 7)], acc[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (!grp) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + (float)fr[i][0];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = (float)(t1 - t0) / iters; out[(blockIdx.x * 8 + wave) * 2 + 1] = s; }
}
int main() {
  float* d; const int nb = 256;
  hipMalloc(&d, nb * 16 * 4);
  unsigned char* src; hipMalloc(&src, 64u * (2u << 20) + (4u << 20)); hipMemset(src, 1, 64u * (2u << 20) + (4u << 20));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = 8; mode <= 9; ++mode)
    for (int pieces : {8, 6, 4}) {
      k<<<nb, 512, 160 * 1024>>>(d, 2000, mode, pieces, src);
      k<<<nb, 512, 160 * 1024>>>(d, 2000, mode, pieces, src);
      float h[nb * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      double a = 0;
      for (int blk = 0; blk < nb; ++blk) for (int w = 0; w < 8; ++w) a += h[(blk * 8 + w) * 2];
      printf("%s, %d of 8 pieces per row (%d KiB per iteration and CU): %.0f cycles per iteration (two phases)\n", mode == 8 ? "DMA | 32 MFMAs" : "DMA alone", pieces, 6 * pieces, a / (nb * 8));
    }
  return 0;
}
