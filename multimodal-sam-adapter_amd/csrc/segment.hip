// Segmentor-side glue on device: what EncoderDecoder does with the head's logits
// (segmentation/mmseg_custom/models/segmentors/encoder_decoder.py):
//   bilinear_accum_nchw : resize(logits, size=crop.shape[2:], 'bilinear', align_corners=False) (ED:90-94) written -- or, for
//                         slide inference, ADDED (preds += F.pad(crop_seg_logit, ...), count_mat[...] += 1, ED:213-219) -- into a
//                         window (y0, x0, hc, wc) of a [B, C, Hd, Wd] canvas;
//   div_count_nchw      : preds / count_mat (ED:225);
//   argmax_nchw         : seg_logit.argmax(dim=1) (ED:477; the softmax of ED:449 is monotonic) -> uint8 class map.
// All are one pass over the canvas, HBM-bound.
#include "common.h"

__global__ __launch_bounds__(256) void bilinear_accum_kernel(const float* __restrict__ src, int C, int hs, int ws, long sstrideB,
                                                             float* __restrict__ dst, int Hd, int Wd, int y0, int x0, int hc, int wc,
                                                             float* __restrict__ count, float rh, float rw, int accumulate) {
  const int j = blockIdx.x * 256 + threadIdx.x;       // column inside the window
  if (j >= wc) return;
  const int i = blockIdx.y % hc, c = blockIdx.y / hc, b = blockIdx.z;
  // PyTorch upsample_bilinear2d, align_corners=False: src = (dst + 0.5) * in/out - 0.5, clamped at 0
  float sh = ((float)i + 0.5f) * rh - 0.5f, sw = ((float)j + 0.5f) * rw - 0.5f;
  sh = sh < 0.f ? 0.f : sh;
  sw = sw < 0.f ? 0.f : sw;
  const int h0 = min((int)sh, hs - 1), w0 = min((int)sw, ws - 1);
  const int h1 = h0 + (h0 < hs - 1 ? 1 : 0), w1 = w0 + (w0 < ws - 1 ? 1 : 0);
  const float lh = sh - (float)h0, lw = sw - (float)w0;
  const float* sp = src + (long)b * sstrideB + (long)c * hs * ws;
  const float v = (1.f - lh) * ((1.f - lw) * sp[h0 * ws + w0] + lw * sp[h0 * ws + w1]) +
                  lh * ((1.f - lw) * sp[h1 * ws + w0] + lw * sp[h1 * ws + w1]);
  const long o = (((long)b * C + c) * Hd + (y0 + i)) * Wd + (x0 + j);
  dst[o] = accumulate ? dst[o] + v : v;
  if (count && c == 0) count[((long)b * Hd + (y0 + i)) * Wd + (x0 + j)] += 1.0f;
}

extern "C" int mmsa_bilinear_accum_nchw(const float* src, long src_strideB, int B, int C, int hs, int ws, float* dst, int Hd, int Wd,
                                        int y0, int x0, int hc, int wc, float* count, int accumulate, hipStream_t stream) {
  MMSA_CHECK_ARG(src && dst && B > 0 && C > 0 && hs > 0 && ws > 0 && hc > 0 && wc > 0, "bilinear_accum_nchw: bad args");
  MMSA_CHECK_ARG(y0 >= 0 && x0 >= 0 && y0 + hc <= Hd && x0 + wc <= Wd, "bilinear_accum_nchw: window (%d,%d)+(%d,%d) outside the %dx%d canvas", y0, x0, hc, wc, Hd, Wd);
  MMSA_CHECK_ARG((long)C * hc <= 65535 && B <= 65535, "bilinear_accum_nchw: C*hc too large for the launch grid");
  dim3 grid(cdiv(wc, 256), C * hc, B);
  hipLaunchKernelGGL(bilinear_accum_kernel, grid, dim3(256), 0, stream, src, C, hs, ws, src_strideB, dst, Hd, Wd, y0, x0, hc, wc, count,
                     (float)hs / (float)hc, (float)ws / (float)wc, accumulate);
  MMSA_CHECK_LAUNCH("bilinear_accum_nchw");
  return MMSA_OK;
}

__global__ __launch_bounds__(256) void div_count_kernel(float* __restrict__ x, const float* __restrict__ count, int C, long HW, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long p = i % HW;
  const long b = i / (HW * C);
  x[i] = x[i] / count[b * HW + p];
}

extern "C" int mmsa_div_count_nchw(float* x, const float* count, int B, int C, long HW, hipStream_t stream) {
  MMSA_CHECK_ARG(x && count && B > 0 && C > 0 && HW > 0, "div_count_nchw: bad args");
  const long total = (long)B * C * HW;
  hipLaunchKernelGGL(div_count_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, count, C, HW, total);
  MMSA_CHECK_LAUNCH("div_count_nchw");
  return MMSA_OK;
}

// first maximum wins, like torch.argmax on ties
__global__ __launch_bounds__(256) void argmax_nchw_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int C, long HW, long total) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // b * HW + p
  if (i >= total) return;
  const long b = i / HW, p = i - b * HW;
  const float* xp = x + b * C * HW + p;
  float best = xp[0];
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    const float v = xp[(long)c * HW];
    if (v > best) { best = v; bi = c; }
  }
  out[i] = (unsigned char)bi;
}

extern "C" int mmsa_argmax_nchw(const float* x, unsigned char* out, int B, int C, long HW, hipStream_t stream) {
  MMSA_CHECK_ARG(x && out && B > 0 && C > 0 && C <= 256 && HW > 0, "argmax_nchw: bad args (C <= 256 for the uint8 map)");
  const long total = (long)B * HW;
  hipLaunchKernelGGL(argmax_nchw_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, out, C, HW, total);
  MMSA_CHECK_LAUNCH("argmax_nchw");
  return MMSA_OK;
}
