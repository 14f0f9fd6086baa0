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
      if (xtok) {   // kernel-uniform; nullptr: add_vit_feature = False (BK:326), the map alone goes through the norm
        const float* xb = xtok + (long)b * Hx * Wx * ldx + c;
        const float v00 = xb[((long)h0 * Wx + w0) * ldx], v01 = xb[((long)h0 * Wx + w1) * ldx];
        const float v10 = xb[((long)h1 * Wx + w0) * ldx], v11 = xb[((long)h1 * Wx + w1) * ldx];
        const float r = (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11);
        v = v + r;
      }
      v = v * bn_scale[c] + bn_shift[c];
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

// The same pass on 64 (pixels) x 64 (channels) tiles with 16-byte accesses on both sides (the 32 x 32 tile above moved 4 bytes per lane and
// reached ~2.5 TB/s: profiles/r02_gemm_traffic.json): a lane reads four consecutive channels of a pixel (c map + the four bilinear taps of
// x as float4), the tile is transposed through LDS (row stride 65 floats: the four scalar writes of a lane and the four scalar reads of the
// store phase are at most 2-way conflicted), and a lane stores four consecutive pixels of a channel (16 lanes = 256 contiguous bytes of an
// NCHW row).  The optional planes output leaves as whole 16-byte chunks through the lane-pair exchange of common.h.
__global__ __launch_bounds__(256) void tail_fuse64_kernel(const float* __restrict__ cmap, long ldc, long cstrideB, const float* __restrict__ xtok, long ldx,
                                                          const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                          float* __restrict__ out, unsigned short* __restrict__ outp, long ldp,
                                                          int Hc, int Wc, int Hx, int Wx, int C, float rh, float rw) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int pix0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  const int cg = t & 15, pr = t >> 4;          // load phase: channels c0 + 4 cg .. + 3 of pixel rows pr, pr + 16, ...
  const int npix = Hc * Wc;
  const int c = c0 + 4 * cg;
  const bool c_ok = c < C;                     // C % 4 == 0: a lane's four channels are in or out together
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sf = sc;
  if (c_ok) { sc = *reinterpret_cast<const float4*>(bn_scale + c); sf = *reinterpret_cast<const float4*>(bn_shift + c); }
  const float* xb = xtok + (long)b * Hx * Wx * ldx + c;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = pr + 16 * k;
    const int pix = pix0 + pl;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool ok = pix < npix && c_ok;
    if (ok) {
      const int h = pix / Wc, w = pix - h * Wc;
      v = *reinterpret_cast<const float4*>(cmap + (long)b * cstrideB + (long)pix * ldc + c);
      float sh = ((float)h + 0.5f) * rh - 0.5f;     // PyTorch upsample_bilinear2d, align_corners=False (as above)
      float sw = ((float)w + 0.5f) * rw - 0.5f;
      sh = sh < 0.f ? 0.f : sh;
      sw = sw < 0.f ? 0.f : sw;
      const int h0 = (int)sh, w0 = (int)sw;
      const int h1 = h0 + (h0 < Hx - 1 ? 1 : 0), w1 = w0 + (w0 < Wx - 1 ? 1 : 0);
      const float lh = sh - (float)h0, lw = sw - (float)w0;
      if (xtok) {   // kernel-uniform; nullptr: add_vit_feature = False (BK:326)
        const float4 v00 = *reinterpret_cast<const float4*>(xb + ((long)h0 * Wx + w0) * ldx), v01 = *reinterpret_cast<const float4*>(xb + ((long)h0 * Wx + w1) * ldx);
        const float4 v10 = *reinterpret_cast<const float4*>(xb + ((long)h1 * Wx + w0) * ldx), v11 = *reinterpret_cast<const float4*>(xb + ((long)h1 * Wx + w1) * ldx);
#define TF_ONE(f_) { const float r_ = (1.f - lh) * ((1.f - lw) * v00.f_ + lw * v01.f_) + lh * ((1.f - lw) * v10.f_ + lw * v11.f_); v.f_ = (v.f_ + r_) * sc.f_ + sf.f_; }
        TF_ONE(x) TF_ONE(y) TF_ONE(z) TF_ONE(w)
#undef TF_ONE
      } else {
        v.x = v.x * sc.x + sf.x; v.y = v.y * sc.y + sf.y; v.z = v.z * sc.z + sf.z; v.w = v.w * sc.w + sf.w;
      }
    }
    if (outp) {   // planes, token-major: the lane pair (cg even | odd) holds 8 consecutive channels = one 16-byte hi chunk + one 16-byte lo chunk
      unsigned short* row = outp + ((long)b * npix + min(pix, npix - 1)) * ldp;
      store_planes8_pair<1>(row, c & ~7, v, MMSA_FMT_B3, (cg & 1) != 0, ok);
    }
    tile[pl][4 * cg + 0] = v.x; tile[pl][4 * cg + 1] = v.y; tile[pl][4 * cg + 2] = v.z; tile[pl][4 * cg + 3] = v.w;
  }
  __syncthreads();
  const int pg = t & 15, cr = t >> 4;          // store phase: pixels pix0 + 4 pg .. + 3 of channel rows cr, cr + 16, ...
  const bool vec = (npix & 3) == 0;            // kernel-uniform: NCHW rows are 16-byte aligned
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = cr + 16 * k;
    const int cc = c0 + cl;
    const int pix = pix0 + 4 * pg;
    if (cc >= C || pix >= npix) continue;
    const float4 o = make_float4(tile[4 * pg][cl], tile[4 * pg + 1][cl], tile[4 * pg + 2][cl], tile[4 * pg + 3][cl]);
    float* dst = out + ((long)b * C + cc) * npix + pix;
    if (vec) *reinterpret_cast<float4*>(dst) = o;
    else {
      dst[0] = o.x;
      if (pix + 1 < npix) dst[1] = o.y;
      if (pix + 2 < npix) dst[2] = o.z;
      if (pix + 3 < npix) dst[3] = o.w;
    }
  }
}

extern "C" int mmsa_tail_fuse(const float* cmap, long ldc, long cstrideB, const float* xtok, long ldx, const float* bn_scale,
                              const float* bn_shift, float* out, unsigned short* out_planes, long ldp, int B, int Hc, int Wc,
                              int Hx, int Wx, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(cmap && bn_scale && bn_shift && out, "tail_fuse: null pointer");   // xtok may be null: no ViT feature added (BK:326)
  MMSA_CHECK_ARG(B > 0 && Hc > 0 && Wc > 0 && Hx > 0 && Wx > 0 && C > 0, "tail_fuse: bad shape");
  MMSA_CHECK_ARG(!out_planes || ((C & 31) == 0 && ldp >= 2L * C), "tail_fuse: planes output needs C % 32 == 0 and ldp >= 2C");
  // scale_factor s = Hc/Hx is what the reference passes (4, 2, 1, 0.5); PyTorch uses 1/s as the source step
  const float rh = (float)Hx / (float)Hc, rw = (float)Wx / (float)Wc;
  const bool old_tiles = MMSA_KNOB("MMSA_TAIL32", 0) != 0;   // A/B aid
  const bool wide = !old_tiles && (C & 3) == 0 && (ldc & 3) == 0 && (!xtok || (ldx & 3) == 0) && (cstrideB & 3) == 0 &&
                    ((((uintptr_t)cmap) | ((uintptr_t)xtok) | ((uintptr_t)bn_scale) | ((uintptr_t)bn_shift) | ((uintptr_t)out)) & 15) == 0 &&
                    (!out_planes || ((C & 7) == 0 && (ldp & 7) == 0 && (((uintptr_t)out_planes) & 15) == 0));
  if (wide) {
    dim3 grid64(cdiv((long)Hc * Wc, 64), cdiv(C, 64), B);
    hipLaunchKernelGGL(tail_fuse64_kernel, grid64, dim3(256), 0, stream, cmap, ldc, cstrideB, xtok, ldx, bn_scale, bn_shift, out, out_planes, ldp, Hc, Wc, Hx, Wx, C, rh, rw);
    MMSA_CHECK_LAUNCH("tail_fuse");
    return MMSA_OK;
  }
  dim3 grid(cdiv((long)Hc * Wc, 32), cdiv(C, 32), B);
  hipLaunchKernelGGL(tail_fuse_kernel, grid, dim3(256), 0, stream, cmap, ldc, cstrideB, xtok, ldx, bn_scale, bn_shift, out, out_planes, ldp, Hc, Wc, Hx, Wx, C, rh, rw);
  MMSA_CHECK_LAUNCH("tail_fuse");
  return MMSA_OK;
}
