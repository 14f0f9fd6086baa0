// Do one wave's ds_read_b128 stream and its SIMD partner's MFMA stream run concurrently?  512-thread workgroups (two waves per SIMD),
// waves 0-3 = group A, 4-7 = group B.  Modes: 0 A reads / B idle, 1 A idle / B MFMA, 2 A reads / B MFMA, 3 both groups alternate
// (A reads while B MFMAs, barrier, swap) -- the ping-pong pattern, 4 = mode 3 without the reads, 5 = mode 3 without the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int nread, int nmfma>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, const unsigned char* src) {
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
  auto reads = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < nread) fr[i] = *reinterpret_cast<const bf16x8*>(smem + off + (i & 3) * 2048 + (i >> 2) * 32768);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto reads_nowait = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) if (i < nread) fr[i] = *reinterpret_cast<const bf16x8*>(smem + off + (i & 3) * 2048 + (i >> 2) * 32768);
  };
  auto mfmas = [&]() {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) if (r * 16 + i < nmfma) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[i & 7], fr[8 + (i & 7)], acc[i], 0, 0, 0);
  };
  // LDS-DMA: 6 pieces of 1 KiB per wave and iteration into a 3-slot ring (48 KiB per slot and iteration over the 8 waves), source
  // rows of 128 B from a per-workgroup 2 MiB window (L2-resident after the first pass)
  const unsigned char* gsrc = src + (size_t)(blockIdx.x & 63) * (2u << 20) + (size_t)wave * 6 * 1024 + lane * 16;
  auto dma = [&](int it) {
    const unsigned char* g = gsrc + (size_t)(it & 31) * 49152;
    unsigned char* l = smem + (it % 3) * 49152 + wave * 6 * 1024;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i * 1024), (__attribute__((address_space(3))) void*)(l + i * 1024), 16, 0, 0);
  };
  long t0 = __builtin_readcyclecounter();
  if (mode <= 2) {
    for (int it = 0; it < iters; ++it) {
      if (grp == 0) { if (mode != 1) reads(); }
      else { if (mode != 0) mfmas(); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (mode <= 5) {
    if (grp) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
      if (mode != 4) reads();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (mode != 5) mfmas();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    if (!grp) __builtin_amdgcn_s_barrier();
  }
  if (mode >= 6) {   // 6: reads + DMA | MFMA, 7: reads + DMA only, 8: DMA | MFMA, 9: DMA only
    if (grp) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
      if (mode <= 7) reads_nowait();
      __builtin_amdgcn_sched_barrier(0);
      dma(it);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (mode == 6 || mode == 8) mfmas();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (!grp) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + (float)fr[i][0];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = (float)(t1 - t0) / iters; out[(blockIdx.x * 8 + wave) * 2 + 1] = s; }
}
template <int nread, int nmfma> void run(float* d, const unsigned char* src);
int main(int argc, char** argv) {
  float* d; const int nb = 256;
  hipMalloc(&d, nb * 16 * 4);
  unsigned char* src; hipMalloc(&src, 64u * (2u << 20) + (4u << 20)); hipMemset(src, 1, 64u * (2u << 20) + (4u << 20));
  run<16, 48>(d, src); run<16, 32>(d, src); run<16, 16>(d, src);
  return 0;
}
template <int nread, int nmfma> void run(float* d, const unsigned char* src) {
  const int nb = 256;
  hipFuncSetAttribute((const void*)k<nread, nmfma>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int mode = 3; mode < 10; ++mode) {
    k<nread, nmfma><<<nb, 512, 160 * 1024>>>(d, 2000, mode, src);
    k<nread, nmfma><<<nb, 512, 160 * 1024>>>(d, 2000, mode, src);
    float h[nb * 16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int blk = 0; blk < nb; ++blk) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[(blk * 8 + w) * 2];
    printf("mode %d (%d reads, %d MFMAs): cycles per iteration group A %.0f group B %.0f\n", mode, nread, nmfma, a / (nb * 4), b / (nb * 4));
  }
}
