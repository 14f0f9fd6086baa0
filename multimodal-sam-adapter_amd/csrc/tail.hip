// Encoder tail (BK:316-337): f_k = BatchNorm_eval( c_k + bilinear_resize(x_k) ), NHWC -> NCHW.
//   c : [B, Hc, Wc, C] token-major (NHWC) fp32;  x : ViT stage output tokens [B, Hx, Wx, C];
//   F.interpolate(mode='bilinear', align_corners=False, scale_factor = Hc/Hx) semantics (BK:328-330);
//   SyncBatchNorm in eval = per-channel affine from running stats (eps 1e-5), folded to scale/shift at pack time.
// The 1/4-resolution map alone is 268 MB fp32 per image, so add + resize + BN + layout change are one pass:
// a 32(pixels) x 32(channels) tile is read channel-contiguous, transposed through LDS (padded, conflict-free)
// and written pixel-contiguous into the NCHW output the mmseg head expects.
#include "common.h"

__global__ __launch_bounds__(256) void tail_fuse_kernel(const float* __restrict__ cmap, long ldc, long cstrideB, const float* __restrict__ xtok, long ldx,
                                                        const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                        float* __restrict__ out, unsigned short* __restrict__ outp, long ldp,
                                                        int Hc, int Wc, int Hx, int Wx, int C, float rh, float rw) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int pix0 = blockIdx.x * 32;          // pixel index within the image (h*Wc + w)
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // ty 0..7
  const int npix = Hc * Wc;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = ty + 8 * k;
    const int pix = pix0 + pl;
    const int c = c0 + tx;
    float v = 0.f;
    if (pix < npix && c < C) {
      const int h = pix / Wc, w = pix - h * Wc;
      v = cmap[(long)b * cstrideB + (long)pix * ldc + c];
      // PyTorch upsample_bilinear2d, align_corners=False: src = (dst + 0.5) * (1/scale) - 0.5, clamped at 0
      float sh = ((float)h + 0.5f) * rh - 0.5f;
      float sw = ((float)w + 0.5f) * rw - 0.5f;
      sh = sh < 0.f ? 0.f : sh;
      sw = sw < 0.f ? 0.f : sw;
      const int h0 = (int)sh, w0 = (int)sw;
      const int h1 = h0 + (h0 < Hx - 1 ? 1 : 0), w1 = w0 + (w0 < Wx - 1 ? 1 : 0);
      const float lh = sh - (float)h0, lw = sw - (float)w0;
      const float* xb = xtok + (long)b * Hx * Wx * ldx + c;
      const float v00 = xb[((long)h0 * Wx + w0) * ldx], v01 = xb[((long)h0 * Wx + w1) * ldx];
      const float v10 = xb[((long)h1 * Wx + w0) * ldx], v11 = xb[((long)h1 * Wx + w1) * ldx];
      const float r = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
      v = (v + r) * bn_scale[c] + bn_shift[c];
      if (outp) {   // the same map token-major as interleaved planes: the A operand of the decode head's first 1x1 conv
        unsigned short hh, ll;
        split_bf16(v, hh, ll);
        unsigned short* q_ = outp + ((long)b * npix + pix) * ldp + ilv(c);
        q_[0] = hh;
        q_[32] = ll;
      }
    }
    tile[pl][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = ty + 8 * k;
    const int c = c0 + cl;
    const int pix = pix0 + tx;
    if (c < C && pix < npix) out[((long)b * C + c) * npix + pix] = tile[tx][cl];
  }
}

extern "C" int mmsa_tail_fuse(const float* cmap, long ldc, long cstrideB, const float* xtok, long ldx, const float* bn_scale,
                              const float* bn_shift, float* out, unsigned short* out_planes, long ldp, int B, int Hc, int Wc,
                              int Hx, int Wx, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(cmap && xtok && bn_scale && bn_shift && out, "tail_fuse: null pointer");
  MMSA_CHECK_ARG(B > 0 && Hc > 0 && Wc > 0 && Hx > 0 && Wx > 0 && C > 0, "tail_fuse: bad shape");
  MMSA_CHECK_ARG(!out_planes || ((C & 31) == 0 && ldp >= 2L * C), "tail_fuse: planes output needs C % 32 == 0 and ldp >= 2C");
  // scale_factor s = Hc/Hx is what the reference passes (4, 2, 1, 0.5); PyTorch uses 1/s as the source step
  const float rh = (float)Hx / (float)Hc, rw = (float)Wx / (float)Wc;
  dim3 grid(cdiv((long)Hc * Wc, 32), cdiv(C, 32), B);
  hipLaunchKernelGGL(tail_fuse_kernel, grid, dim3(256), 0, stream, cmap, ldc, cstrideB, xtok, ldx, bn_scale, bn_shift, out, out_planes, ldp, Hc, Wc, Hx, Wx, C, rh, rw);
  MMSA_CHECK_LAUNCH("tail_fuse");
  return MMSA_OK;
}
