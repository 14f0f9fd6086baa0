// Segmentor-side glue on device: what EncoderDecoder does with the head's logits
// (segmentation/mmseg_custom/models/segmentors/encoder_decoder.py):
//   bilinear_accum_nchw : resize(logits, size=crop.shape[2:], 'bilinear', align_corners=False) (ED:90-94) written -- or, for
//                         slide inference, ADDED (preds += F.pad(crop_seg_logit, ...), count_mat[...] += 1, ED:213-219) -- into a
//                         window (y0, x0, hc, wc) of a [B, C, Hd, Wd] canvas;
//   div_count_nchw      : preds / count_mat (ED:225);
//   argmax_nchw         : seg_logit.argmax(dim=1) (ED:477; the softmax of ED:449 is monotonic) -> uint8 class map.
// All are one pass over the canvas, HBM-bound.
#include "common.h"

// No fused multiply-add contraction in this file: the canvas path (bilinear_accum + div_count + argmax) and the one-pass class map
// (slide_argmax) must round the SAME interpolation formula identically, or an exact tie between two classes in one of them is not a
// tie in the other (seen once in 2 073 600 pixels of a 1080 x 1920 frame, where the compiler had contracted the two differently).
#pragma clang fp contract(off)

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

// ---- crop extraction of slide inference (ED:205-212: crop_img = img[:, :, y1:y2, x1:x2]) as one launch for a batch of windows:
// dst[k] = src[b_k, :, y0_k : y0_k + hc, x0_k : x0_k + wc].  The window table travels by value in the launch arguments.
#define MMSA_MAX_WINDOWS 64
struct WindowTable { int n; int b[MMSA_MAX_WINDOWS], y0[MMSA_MAX_WINDOWS], x0[MMSA_MAX_WINDOWS]; };

__global__ __launch_bounds__(256) void crop_batch_kernel(const float* __restrict__ src, int C, int H, int W, float* __restrict__ dst,
                                                         int hc, int wc, WindowTable wt) {
  const int k = blockIdx.z, c = blockIdx.y / hc, i = blockIdx.y - c * hc;
  const float* s = src + (((long)wt.b[k] * C + c) * H + (wt.y0[k] + i)) * W + wt.x0[k];
  float* d = dst + (((long)k * C + c) * hc + i) * wc;
  for (int j = threadIdx.x; j < wc; j += 256) d[j] = s[j];
}

static int fill_windows(WindowTable& wt, const int* windows, int n, int B, int H, int W, int hc, int wc, const char* name) {
  MMSA_CHECK_ARG(windows && n > 0 && n <= MMSA_MAX_WINDOWS, "%s: 1..%d windows per call", name, MMSA_MAX_WINDOWS);
  wt.n = n;
  for (int k = 0; k < n; ++k) {
    wt.b[k] = windows[3 * k]; wt.y0[k] = windows[3 * k + 1]; wt.x0[k] = windows[3 * k + 2];
    MMSA_CHECK_ARG(wt.b[k] >= 0 && wt.b[k] < B && wt.y0[k] >= 0 && wt.x0[k] >= 0 && wt.y0[k] + hc <= H && wt.x0[k] + wc <= W,
                   "%s: window %d (image %d, y0 %d, x0 %d, %dx%d) outside the [%d, %d, %d] input", name, k, wt.b[k], wt.y0[k], wt.x0[k], hc, wc, B, H, W);
  }
  return MMSA_OK;
}

extern "C" int mmsa_crop_batch_nchw(const float* src, int B, int C, int H, int W, const int* windows /* HOST [n,3]: image, y0, x0 */, int n,
                                    float* dst, int hc, int wc, hipStream_t stream) {
  MMSA_CHECK_ARG(src && dst && B > 0 && C > 0 && hc > 0 && wc > 0 && (long)C * hc <= 65535, "crop_batch_nchw: bad args");
  WindowTable wt;
  int rc = fill_windows(wt, windows, n, B, H, W, hc, wc, "crop_batch_nchw");
  if (rc) return rc;
  hipLaunchKernelGGL(crop_batch_kernel, dim3(1, C * hc, n), dim3(256), 0, stream, src, C, H, W, dst, hc, wc, wt);
  MMSA_CHECK_LAUNCH("crop_batch_nchw");
  return MMSA_OK;
}

// ---- class map of a whole sliding-window frame in ONE pass (ED:213-225 + ED:449,477 fused): for every pixel of image b,
//   preds[c] = sum over the windows k of image b that cover it, IN WINDOW ORDER, of bilinear_{align_corners=False}(logits_k -> hc x wc)[c]
//   out      = argmax_c preds[c] / count          (first maximum wins; count = number of covering windows, ED:219,225)
// -- the same additions in the same order as bilinear_accum (accumulate) + div_count + argmax, without the [B, C, H, W] fp32 canvas
// (207 MB for a 1080 x 1920 frame and 25 classes, written and re-read once per window).  With one full-size window per image it is
// the whole-image `resize x4 + argmax` of ED:90-94,477.  Pixels no window covers make the call fail (ED:220 asserts the same).
__global__ __launch_bounds__(256) void slide_argmax_kernel(const float* __restrict__ logits, int C, int hs, int ws, unsigned char* __restrict__ out,
                                                           int H, int W, int hc, int wc, float rh, float rw, WindowTable wt, int* __restrict__ uncovered) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  // covering windows (at most 8 per pixel) and their 4-tap coordinates (PyTorch upsample_bilinear2d: src = (dst + 0.5) * in/out - 0.5,
  // clamped at 0).  The slot arrays are only ever indexed by unrolled constants (predicated inserts), so they live in registers.
  int nk = 0, kk[8], o00[8], o01[8], o10[8], o11[8];
  float lhs[8], lws[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { kk[q] = 0; o00[q] = o01[q] = o10[q] = o11[q] = 0; lhs[q] = lws[q] = 0.f; }
  for (int k = 0; k < wt.n; ++k) {
    if (wt.b[k] != b) continue;
    const int i = y - wt.y0[k], j = x - wt.x0[k];
    if (i < 0 || i >= hc || j < 0 || j >= wc) continue;
    float sh = ((float)i + 0.5f) * rh - 0.5f, sw = ((float)j + 0.5f) * rw - 0.5f;
    sh = sh < 0.f ? 0.f : sh;
    sw = sw < 0.f ? 0.f : sw;
    const int h0 = min((int)sh, hs - 1), w0 = min((int)sw, ws - 1);
    const int h1 = h0 + (h0 < hs - 1 ? 1 : 0), w1 = w0 + (w0 < ws - 1 ? 1 : 0);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q == nk) {
        kk[q] = k; lhs[q] = sh - (float)h0; lws[q] = sw - (float)w0;
        o00[q] = h0 * ws + w0; o01[q] = h0 * ws + w1; o10[q] = h1 * ws + w0; o11[q] = h1 * ws + w1;
      }
    ++nk;
  }
  if (nk == 0 || nk > 8) {   // no window, or more than 8 overlapping windows per pixel: not supported -- counted, and the pixel gets 255, never an unwritten byte
    atomicAdd(uncovered, 1);
    out[((long)b * H + y) * W + x] = 255;
    return;
  }
  const float cnt = (float)nk;
  float best = -INFINITY;
  int bi = 0;
  for (int c = 0; c < C; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q < nk) {
        const float* sp = logits + ((long)kk[q] * C + c) * hs * ws;
        const float lh = lhs[q], lw = lws[q];
        const float v = (1.f - lh) * ((1.f - lw) * sp[o00[q]] + lw * sp[o01[q]]) + lh * ((1.f - lw) * sp[o10[q]] + lw * sp[o11[q]]);
        acc = q == 0 ? v : acc + v;     // window order: the first window WRITES (0 + v == v), later ones add
      }
    }
    const float p = acc / cnt;
    if (c == 0 || p > best) { best = p; bi = c; }
  }
  out[((long)b * H + y) * W + x] = (unsigned char)bi;
}

extern "C" int mmsa_slide_argmax(const float* logits, int n, int C, int hs, int ws, const int* windows /* HOST [n,3] */, unsigned char* out,
                                 int B, int H, int W, int hc, int wc, int* uncovered /* device int, zeroed by the caller */, hipStream_t stream) {
  MMSA_CHECK_ARG(logits && out && uncovered && C > 0 && C <= 255 && hs > 0 && ws > 0 && hc > 0 && wc > 0 && B > 0 && H <= 65535 && B <= 65535, "slide_argmax: bad args");
  WindowTable wt;
  int rc = fill_windows(wt, windows, n, B, H, W, hc, wc, "slide_argmax");
  if (rc) return rc;
  hipLaunchKernelGGL(slide_argmax_kernel, dim3(cdiv(W, 256), H, B), dim3(256), 0, stream, logits, C, hs, ws, out, H, W, hc, wc,
                     (float)hs / (float)hc, (float)ws / (float)wc, wt, uncovered);
  MMSA_CHECK_LAUNCH("slide_argmax");
  return MMSA_OK;
}
