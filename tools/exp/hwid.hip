// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 1) void k(unsigned* out) {
  extern __shared__ unsigned char smem[];
  if (threadIdx.x == 9999) smem[0] = 1;
  unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; const int nb = 512;
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<<<nb, 512, 160 * 1024>>>(d);
  unsigned h[nb * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 4; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf(" w%d->simd%u(slot %u)", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15); printf("\n"); }
  return 0;
}
