// ConvNeXt block front half fused: 7x7 depthwise conv (TC:69-70,102) + the block's LayerNorm over channels (TC:103-106,
// mmpretrain_custom/models/utils/norm.py:51-90, eps 1e-6) -> interleaved planes, the A operand of pointwise_conv1.
// The fp32 conv output never goes to memory (the two-kernel form writes it, 25 MB at stage 2, and reads it back).
//
// One workgroup = 8 x 8 output pixels x ALL channels, walked in chunks of 64 channels: the conv outputs of every chunk stay in
// registers (NCH x 4 float4 per lane), so the LayerNorm sees whole channel vectors without a second pass over memory.
//   * per chunk the 14 x 14 x 64 input halo goes through LDS (50 KiB, channel-contiguous: a 16-lane group reads 256 contiguous
//     bytes), DOUBLE-buffered: the loads of chunk k+1 are issued before chunk k is computed and stored to the other buffer
//     after it -- one workgroup per CU (4 waves) has no other wave to hide a load behind;
//   * lane = (4-channel vector cv = tid & 15, 1 x 4 pixel strip = tid >> 4): per kernel row 10 LDS reads serve 7 taps x 4
//     outputs; the chunk's 49 x 64 tap-major weights are staged in LDS with the halo (the 16 strips read them by broadcast);
//   * the 16 lanes that share a strip are one DPP row: the LayerNorm sums (mean, then centred second moment) are 4 xor-shuffles,
//     no LDS, no barrier, fixed order (bit-identical run to run); biased variance, per-stream weights.
// Measured (ViT-L step, stage 2, C = 384, 4 images of 64 x 64): 42.6 us against 31 + 26 us for the kernel pair, 73 against ~80 us at
// C = 192 -- but neither the ConvNeXt chain alone (8.10 vs 8.13 ms, tools/spm_time.py) nor the step (37.7 ms) gets faster: in the
// replayed graph the tail of one small kernel overlaps the ramp of the next, which one 122-KiB workgroup per CU cannot.  Hence off by default
// (backbone.fuse_dwconv_ln); kept as a tested operator.  History: one 4 x 4 x all-channel tile per workgroup (154 KiB halo, 6.25 x
// input re-reads) took 64.6 us; this layout with the tap weights fetched from L2 per kernel row took 79 us (one L2 round trip per
// row and wave, nothing to hide it behind at one wave per SIMD) -- staging them in LDS gave 42.6 us.
#include "common.h"

template <int NCH>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float eps,
                                                         unsigned short* __restrict__ yp, long ldp, long pstrideB,
                                                         int H, int W, int tilesX, int imgs_per_group) {
  constexpr int TW = 14, CB = 64, C = NCH * CB;
  constexpr int NIT = (TW * TW * (CB / 4) + 255) / 256;   // 13 float4 per lane and chunk
  constexpr int HALO = TW * TW * CB, WCH = 49 * CB, STAGE = HALO + WCH;   // floats per stage: halo 12544 + weights 3136
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 x ([14][14][64] halo + [49][64] tap-major weights)
  const int b = blockIdx.z;
  if (imgs_per_group > 0) {   // image groups (the two ConvNeXt streams stacked along the batch) with their own weights
    const int grp = b / imgs_per_group;
    w += (long)grp * 49 * C;
    if (bias) bias += (long)grp * C;
    lnw += (long)grp * C;
    lnb += (long)grp * C;
  }
  const int tx0 = (blockIdx.x % tilesX) * 8, ty0 = (blockIdx.x / tilesX) * 8;
  const float* xb = x + (long)b * xstrideB;
  const int cv = threadIdx.x & 15, strip = threadIdx.x >> 4;   // 16 strips: row = strip >> 1, x0 = (strip & 1) * 4
  const int oy = strip >> 1, ox0 = (strip & 1) * 4;

  float4 hv[NIT], wv[4];
#define HALO_LOAD(k_)                                                                                         \
  _Pragma("unroll") for (int it = 0; it < 4; ++it) {   /* the chunk's 49 x 16 weight vectors */               \
    const int i = threadIdx.x + it * 256;                                                                     \
    wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
    if (i < 49 * 16) wv[it] = *reinterpret_cast<const float4*>(w + (long)(i >> 4) * C + (k_) * CB + (i & 15) * 4); \
  }                                                                                                           \
  _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                        \
    const int i = threadIdx.x + it * 256;                                                                     \
    const int pos = i >> 4;                                                                                   \
    const int ly = pos / TW, lx = pos - ly * TW;                                                              \
    const int iy = ty0 + ly - 3, ix = tx0 + lx - 3;                                                           \
    hv[it] = make_float4(0.f, 0.f, 0.f, 0.f);                                                                 \
    if (i < TW * TW * (CB / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W)                                     \
      hv[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + (k_) * CB + (i & 15) * 4);   \
  }
#define HALO_STORE(buf_)                                                                                      \
  _Pragma("unroll") for (int it = 0; it < 4; ++it) {                                                          \
    const int i = threadIdx.x + it * 256;                                                                     \
    if (i < 49 * 16) *reinterpret_cast<float4*>((buf_) + HALO + i * 4) = wv[it];                              \
  }                                                                                                           \
  _Pragma("unroll") for (int it = 0; it < NIT; ++it) {                                                        \
    const int i = threadIdx.x + it * 256;                                                                     \
    if (i < TW * TW * (CB / 4)) *reinterpret_cast<float4*>((buf_) + (i >> 4) * CB + (i & 15) * 4) = hv[it];   \
  }

  float4 acc[NCH][4];
  HALO_LOAD(0)
  HALO_STORE(lds)
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const float* tile = lds + (k & 1) * STAGE;
    const float* wl = tile + HALO + cv * 4;   // this lane's weight column; the 16 strips read the same addresses (broadcast)
    if (k + 1 < NCH) { HALO_LOAD(k + 1) }
    const int c = k * CB + cv * 4;
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 4; ++p) acc[k][p] = bv;
#pragma unroll 1
    for (int kh = 0; kh < 7; ++kh) {   // weights from LDS: a global load per kernel row cost one L2 round trip per row and wave
      float4 in[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) in[i] = *reinterpret_cast<const float4*>(tile + ((oy + kh) * TW + ox0 + i) * CB + cv * 4);
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const float4 f = *reinterpret_cast<const float4*>(wl + (kh * 7 + kw) * CB);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          acc[k][p].x += in[p + kw].x * f.x; acc[k][p].y += in[p + kw].y * f.y;
          acc[k][p].z += in[p + kw].z * f.z; acc[k][p].w += in[p + kw].w * f.w;
        }
      }
    }
    if (k + 1 < NCH) {
      HALO_STORE(lds + ((k + 1) & 1) * STAGE)   // buffer (k+1)&1 was last read for chunk k-1: every wave passed the barrier below since
      __syncthreads();
    }
  }
#undef HALO_LOAD
#undef HALO_STORE
  // ---- LayerNorm over the C channels of each of this lane's 4 pixels: 16 lanes (cv) x NCH chunks x 4 channels
  float mean[4], rstd[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) s += (acc[k][p].x + acc[k][p].y) + (acc[k][p].z + acc[k][p].w);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64);
    mean[p] = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const float dx = acc[k][p].x - mean[p], dy = acc[k][p].y - mean[p], dz = acc[k][p].z - mean[p], dw = acc[k][p].w - mean[p];
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64); q += __shfl_xor(q, 8, 64);
    rstd[p] = rsqrtf(q / (float)C + eps);
  }
  const int gy = ty0 + oy;
  if (gy >= H) return;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = k * CB + cv * 4;
    const float4 gw = *reinterpret_cast<const float4*>(lnw + c), gb = *reinterpret_cast<const float4*>(lnb + c);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int gx = tx0 + ox0 + p;
      if (gx >= W) continue;
      float4 o;
      o.x = (acc[k][p].x - mean[p]) * rstd[p] * gw.x + gb.x;
      o.y = (acc[k][p].y - mean[p]) * rstd[p] * gw.y + gb.y;
      o.z = (acc[k][p].z - mean[p]) * rstd[p] * gw.z + gb.z;
      o.w = (acc[k][p].w - mean[p]) * rstd[p] * gw.w + gb.w;
      uint2 h2, l2;
      split4(o, h2, l2);
      unsigned short* q_ = yp + (long)b * pstrideB + ((long)gy * W + gx) * ldp + ilv(c);
      *reinterpret_cast<uint2*>(q_) = h2;
      *reinterpret_cast<uint2*>(q_ + 32) = l2;
    }
  }
}

template <int NCH>
static int launch_dwconv7_ln(const float* x, long ldx, long xstrideB, const float* w, const float* bias, const float* lnw,
                             const float* lnb, float eps, unsigned short* yp, long ldp, long pstrideB, int B, int H, int W,
                             int imgs_per_group, hipStream_t stream) {
  const size_t smem = 2 * (14 * 14 * 64 + 49 * 64) * sizeof(float);   // 122.5 KiB
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)dwconv7_ln_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const int tx = cdiv(W, 8), ty = cdiv(H, 8);
  hipLaunchKernelGGL((dwconv7_ln_kernel<NCH>), dim3(tx * ty, 1, B), dim3(256), smem, stream, x, ldx, xstrideB, w, bias, lnw, lnb, eps,
                     yp, ldp, pstrideB, H, W, tx, imgs_per_group);
  MMSA_CHECK_LAUNCH("dwconv7_ln");
  return MMSA_OK;
}

extern "C" int mmsa_dwconv7_ln(const float* x, long ldx, long xstrideB, const float* w, const float* bias, const float* lnw,
                               const float* lnb, float eps, unsigned short* yp, long ldp, long pstrideB, int B, int H, int W, int C,
                               int imgs_per_group, hipStream_t stream) {
  MMSA_CHECK_ARG(x && w && lnw && lnb && yp && B > 0 && H > 0 && W > 0, "dwconv7_ln: bad args");
  MMSA_CHECK_ARG(C >= 64 && C <= 384 && (C & 63) == 0, "dwconv7_ln: C must be a multiple of 64 in 64..384 (use dwconv_nhwc + layernorm_rows otherwise), got %d", C);
  MMSA_CHECK_ARG(imgs_per_group >= 0 && (imgs_per_group == 0 || B % imgs_per_group == 0), "dwconv7_ln: bad image grouping");
  MMSA_CHECK_ARG((ldx & 3) == 0 && (xstrideB & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)lnw) | ((uintptr_t)lnb)) & 15) == 0,
                 "dwconv7_ln: operands must be 16-byte aligned");
  MMSA_CHECK_ARG((((uintptr_t)yp) & 127) == 0 && (ldp & 63) == 0 && (pstrideB & 63) == 0 && ldp >= 2L * C, "dwconv7_ln: bad output planes");
#define DWLN(N_) case N_: return launch_dwconv7_ln<N_>(x, ldx, xstrideB, w, bias, lnw, lnb, eps, yp, ldp, pstrideB, B, H, W, imgs_per_group, stream);
  switch (C / 64) { DWLN(1) DWLN(2) DWLN(3) DWLN(4) DWLN(5) DWLN(6) }
#undef DWLN
  mmsa_set_error("dwconv7_ln: unsupported channel count %d", C);
  return MMSA_ERR_ARG;
}
