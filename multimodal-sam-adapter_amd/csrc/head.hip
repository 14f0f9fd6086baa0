// Segformer all-MLP decode head on device (segformer_head.py:47-66 + mmseg BaseDecodeHead.cls_seg), SURVEY 8(f1).
//
// The head consumes the backbone's NCHW fp32 maps f1..f4 (that is the plugin boundary), runs four 1x1 conv + BN +
// ReLU branches, bilinearly resizes them to the 1/4-resolution grid, concatenates, runs the 1x1 fusion conv + BN +
// ReLU and the 1x1 classifier.  1x1 convolutions are token GEMMs, so the only head-specific kernels are layout moves:
//   nchw_to_planes  : [B,C,HW] fp32 -> [B*HW, C] interleaved bf16 hi/lo planes (the GEMM's A operand), LDS transpose
//   head_fuse       : z0 + sum_i bilinear(z_i) -> BN affine -> ReLU -> planes.  The fusion conv is linear and acts on
//                     channels only, so it commutes with the (spatial, linear) resize and with the concat: it is
//                     applied per branch at the branch's native resolution (4x fewer FLOPs than on the concatenated
//                     1/4-resolution map) and the resized partial products are summed here.
//   tokens_to_nchw  : [B*HW, ld] fp32 -> [B, C, HW] fp32 logits
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void nchw_to_planes_kernel(const float* __restrict__ src, long strideB, int C, long HW,
                                                             uint16_t* __restrict__ planes, long ldp) {
  // tile: one 32-channel k-block x 256 pixels.  row stride 257 floats: transposed reads hit distinct banks.
  __shared__ float tile[32][257];
  const int b = blockIdx.z;
  const long pix0 = (long)blockIdx.x * 256;
  const int c0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* sb = src + (long)b * strideB;
  if (VEC) {   // HW % 4 == 0 and 16-byte aligned rows: one 16-byte load per lane = 1 KiB per wave instruction
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + wv * 8 + j;
      const long pix = pix0 + 4 * lane;
      v[j] = (c < C && pix < HW) ? *reinterpret_cast<const float4*>(sb + (long)c * HW + pix) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float* t = &tile[wv * 8 + j][4 * lane];
      t[0] = v[j].x; t[1] = v[j].y; t[2] = v[j].z; t[3] = v[j].w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int cl = wv * 8 + j;
      const int c = c0 + cl;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long pix = pix0 + lane + 64 * k;
        tile[cl][lane + 64 * k] = (c < C && pix < HW) ? sb[(long)c * HW + pix] : 0.f;
      }
    }
  }
  __syncthreads();
  const int q = threadIdx.x & 3;         // channel octet of the k-block
  const int pl = threadIdx.x >> 2;       // 0..63
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = pl + 64 * k;
    const long pix = pix0 + p;
    if (pix >= HW) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[8 * q + j][p];
    uint4 hi, lo;
    split2(v[0], v[1], hi.x, lo.x);
    split2(v[2], v[3], hi.y, lo.y);
    split2(v[4], v[5], hi.z, lo.z);
    split2(v[6], v[7], hi.w, lo.w);
    uint16_t* row = planes + ((long)b * HW + pix) * ldp + (long)blockIdx.y * 64 + 8 * q;
    *reinterpret_cast<uint4*>(row) = hi;
    *reinterpret_cast<uint4*>(row + 32) = lo;
  }
}

extern "C" int mmsa_nchw_to_planes(const float* src, long strideB, uint16_t* planes, long ldp, int B, int C, long HW,
                                   hipStream_t stream) {
  MMSA_CHECK_ARG(src && planes, "nchw_to_planes: null pointer");
  MMSA_CHECK_ARG(B > 0 && C > 0 && HW > 0, "nchw_to_planes: bad shape");
  const int cpad = (C + 31) / 32 * 32;
  MMSA_CHECK_ARG(ldp >= 2L * cpad && ldp % 64 == 0, "nchw_to_planes: planes row stride %ld < 2*%d or not a multiple of 64", ldp, cpad);
  MMSA_CHECK_ARG((reinterpret_cast<uintptr_t>(planes) & 127) == 0, "nchw_to_planes: planes must be 128-byte aligned");
  MMSA_CHECK_ARG(strideB >= (long)C * HW, "nchw_to_planes: image stride %ld < C*HW", strideB);
  dim3 grid(cdiv(HW, 256), cpad / 32, B);
  if ((HW & 3) == 0 && (strideB & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
    hipLaunchKernelGGL(nchw_to_planes_kernel<true>, grid, dim3(256), 0, stream, src, strideB, C, HW, planes, ldp);
  else
    hipLaunchKernelGGL(nchw_to_planes_kernel<false>, grid, dim3(256), 0, stream, src, strideB, C, HW, planes, ldp);
  MMSA_CHECK_LAUNCH("nchw_to_planes");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
struct HeadLevels {
  const float* z[3];
  int H[3], W[3];
  float rh[3], rw[3];
};

// One wave per output pixel-row chunk: lane handles 8 consecutive channels (C % 8 == 0), loops over C in steps of 512.
__global__ __launch_bounds__(256) void head_fuse_kernel(const float* __restrict__ z0, long ld, HeadLevels lv, int nlv,
                                                        const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                        uint16_t* __restrict__ planes, long ldp, float* __restrict__ out32, long ldo,
                                                        int H, int W, int C, long npix_total, int act) {
  const long pixg = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // b*H*W + h*W + w
  if (pixg >= npix_total) return;
  const int lane = threadIdx.x & 63;
  const unsigned hw = (unsigned)H * (unsigned)W;   // 32-bit index arithmetic (npix_total < 2^31, checked by the launcher)
  const unsigned pg = (unsigned)pixg;
  const int b = (int)(pg / hw);
  const int rem = (int)(pg - (unsigned)b * hw);
  const int h = rem / W, w = rem - h * W;
  // per-level taps (wave-uniform)
  long o00[3], o01[3], o10[3], o11[3];
  float lh[3], lw[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (i < nlv) {
      // F.interpolate(mode='bilinear', align_corners=False): src = (dst + 0.5) * in/out - 0.5, clamped at 0
      float sh = ((float)h + 0.5f) * lv.rh[i] - 0.5f;
      float sw = ((float)w + 0.5f) * lv.rw[i] - 0.5f;
      sh = sh < 0.f ? 0.f : sh;
      sw = sw < 0.f ? 0.f : sw;
      const int h0 = min((int)sh, lv.H[i] - 1), w0 = min((int)sw, lv.W[i] - 1);
      const int h1 = h0 + (h0 < lv.H[i] - 1 ? 1 : 0), w1 = w0 + (w0 < lv.W[i] - 1 ? 1 : 0);
      lh[i] = sh - (float)h0;
      lw[i] = sw - (float)w0;
      const long base = (long)b * lv.H[i] * lv.W[i];
      o00[i] = (base + (long)h0 * lv.W[i] + w0) * ld;
      o01[i] = (base + (long)h0 * lv.W[i] + w1) * ld;
      o10[i] = (base + (long)h1 * lv.W[i] + w0) * ld;
      o11[i] = (base + (long)h1 * lv.W[i] + w1) * ld;
    } else {
      o00[i] = o01[i] = o10[i] = o11[i] = 0; lh[i] = lw[i] = 0.f;
    }
  }
  for (int c = lane * 8; c < C; c += 512) {
    float v[8];
    // every load of the pass first (2 + 24 + 4 vectors; a level beyond nlv re-reads level 0's taps and is not added): the per-level `if`
    // around the loads made each level a separate, fully awaited batch
    float4 t[3][2][4];
    const float4 a = *reinterpret_cast<const float4*>(z0 + pixg * ld + c);
    const float4 d = *reinterpret_cast<const float4*>(z0 + pixg * ld + c + 4);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int il = i < nlv ? i : 0;
      const float* zi = (nlv > 0 ? lv.z[il] : z0 + pixg * ld) + c;   // (no level at all: any valid address)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        t[i][hh][0] = *reinterpret_cast<const float4*>(zi + o00[il] + 4 * hh);
        t[i][hh][1] = *reinterpret_cast<const float4*>(zi + o01[il] + 4 * hh);
        t[i][hh][2] = *reinterpret_cast<const float4*>(zi + o10[il] + 4 * hh);
        t[i][hh][3] = *reinterpret_cast<const float4*>(zi + o11[il] + 4 * hh);
      }
    }
    const float4 s0 = *reinterpret_cast<const float4*>(bn_scale + c), s1 = *reinterpret_cast<const float4*>(bn_scale + c + 4);
    const float4 f0 = *reinterpret_cast<const float4*>(bn_shift + c), f1 = *reinterpret_cast<const float4*>(bn_shift + c + 4);
    __builtin_amdgcn_sched_barrier(0);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < nlv) {
        const float w00 = (1.f - lh[i]) * (1.f - lw[i]), w01 = (1.f - lh[i]) * lw[i], w10 = lh[i] * (1.f - lw[i]), w11 = lh[i] * lw[i];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float4 t00 = t[i][hh][0], t01 = t[i][hh][1], t10 = t[i][hh][2], t11 = t[i][hh][3];
          v[4 * hh + 0] += w00 * t00.x + w01 * t01.x + w10 * t10.x + w11 * t11.x;
          v[4 * hh + 1] += w00 * t00.y + w01 * t01.y + w10 * t10.y + w11 * t11.y;
          v[4 * hh + 2] += w00 * t00.z + w01 * t01.z + w10 * t10.z + w11 * t11.z;
          v[4 * hh + 3] += w00 * t00.w + w01 * t01.w + w10 * t10.w + w11 * t11.w;
        }
      }
    }
    const float bs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, bf[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = apply_act(v[j] * bs[j] + bf[j], act);
    if (out32) {
      *reinterpret_cast<float4*>(out32 + pixg * ldo + c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out32 + pixg * ldo + c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (planes) {
      uint4 hi, lo;
      split2(v[0], v[1], hi.x, lo.x);
      split2(v[2], v[3], hi.y, lo.y);
      split2(v[4], v[5], hi.z, lo.z);
      split2(v[6], v[7], hi.w, lo.w);
      uint16_t* row = planes + pixg * ldp + ilv(c);
      *reinterpret_cast<uint4*>(row) = hi;
      *reinterpret_cast<uint4*>(row + 32) = lo;
    }
  }
}

extern "C" int mmsa_head_fuse(const float* z0, const float* z1, int H1, int W1, const float* z2, int H2, int W2, const float* z3,
                              int H3, int W3, long ld, const float* bn_scale, const float* bn_shift, uint16_t* planes, long ldp,
                              float* out32, long ldo, int B, int H, int W, int C, int act, hipStream_t stream) {
  MMSA_CHECK_ARG(z0 && bn_scale && bn_shift && (planes || out32), "head_fuse: null pointer");
  MMSA_CHECK_ARG(((((uintptr_t)bn_scale) | ((uintptr_t)bn_shift) | ((uintptr_t)z0)) & 15) == 0, "head_fuse: z0 / bn_scale / bn_shift must be 16-byte aligned");
  MMSA_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "head_fuse: bad shape (C=%d must be a multiple of 8)", C);
  MMSA_CHECK_ARG(ld >= C && ld % 4 == 0, "head_fuse: row stride %ld", ld);
  MMSA_CHECK_ARG(!planes || (ldp >= 2L * ((C + 31) / 32 * 32) && ldp % 64 == 0 && (reinterpret_cast<uintptr_t>(planes) & 127) == 0),
                 "head_fuse: planes stride/alignment");
  MMSA_CHECK_ARG(!out32 || (ldo >= C && ldo % 4 == 0), "head_fuse: fp32 out stride %ld", ldo);
  HeadLevels lv;
  int n = 0;
  const float* zs[3] = {z1, z2, z3};
  const int hs[3] = {H1, H2, H3}, wsz[3] = {W1, W2, W3};
  for (int i = 0; i < 3; ++i) {
    lv.z[i] = nullptr; lv.H[i] = lv.W[i] = 1; lv.rh[i] = lv.rw[i] = 1.f;
  }
  for (int i = 0; i < 3; ++i) {
    if (!zs[i]) continue;
    MMSA_CHECK_ARG(hs[i] > 0 && wsz[i] > 0, "head_fuse: level %d has bad size", i + 1);
    lv.z[n] = zs[i]; lv.H[n] = hs[i]; lv.W[n] = wsz[i];
    lv.rh[n] = (float)hs[i] / (float)H;
    lv.rw[n] = (float)wsz[i] / (float)W;
    ++n;
  }
  const long npix = (long)B * H * W;
  MMSA_CHECK_ARG(npix < (1L << 31), "head_fuse: too many pixels for the 32-bit index arithmetic");
  hipLaunchKernelGGL(head_fuse_kernel, dim3(cdiv(npix, 4)), dim3(256), 0, stream, z0, ld, lv, n, bn_scale, bn_shift, planes, ldp,
                     out32, ldo, H, W, C, npix, act);
  MMSA_CHECK_LAUNCH("head_fuse");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tokens_to_nchw_kernel(const float* __restrict__ src, long ld, float* __restrict__ dst, long HW, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long pix0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pl = ty + 8 * k;
    const long pix = pix0 + pl;
    const int c = c0 + tx;
    tile[pl][tx] = (pix < HW && c < C) ? src[((long)b * HW + pix) * ld + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k;
    const long pix = pix0 + tx;
    if (c < C && pix < HW) dst[((long)b * C + c) * HW + pix] = tile[tx][ty + 8 * k];
  }
}

extern "C" int mmsa_tokens_to_nchw(const float* src, long ld, float* dst, int B, long HW, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(src && dst, "tokens_to_nchw: null pointer");
  MMSA_CHECK_ARG(B > 0 && HW > 0 && C > 0 && ld >= C, "tokens_to_nchw: bad shape");
  dim3 grid(cdiv(HW, 32), cdiv(C, 32), B);
  hipLaunchKernelGGL(tokens_to_nchw_kernel, grid, dim3(256), 0, stream, src, ld, dst, HW, C);
  MMSA_CHECK_LAUNCH("tokens_to_nchw");
  return MMSA_OK;
}
