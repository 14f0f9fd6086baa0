#include <hip/hip_runtime.h>
#define KNAME pair_slp_on
#include "pair_kernel.inc"
extern "C" void launch_slp_on(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int C, hipStream_t s) {
  hipLaunchKernelGGL(pair_slp_on, dim3((W * (C >> 2) + 255) / 256, B * H), dim3(256), 0, s, x, ldx, w, y, ldy, H, W, C);
}
