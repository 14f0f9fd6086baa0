// ConvNeXt block front half fused: 7x7 depthwise conv (TC:69-70,102) + the block's LayerNorm over channels (TC:103-106,
// mmpretrain_custom/models/utils/norm.py:51-90, eps 1e-6) -> interleaved planes, the A operand of pointwise_conv1.
//
// One workgroup = TY x TX output pixels x ALL C channels (LayerNorm needs the whole channel vector of a pixel), so the
// fp32 conv output never goes to memory (the two-kernel form writes it, 25 MB at stage 2, and reads it back for the LayerNorm).
// MEASURED SLOWER than the pair and therefore NOT on the default path (backbone.fuse_dwconv_ln = False): 64.6 us against
// 31 + 26 us at C = 384, 114 against ~80 at C = 192, 172 against 168 at C = 96.  All channels of a pixel in one workgroup
// forces a 4 x 4 tile at C = 384: its 10 x 10 halo re-reads the input 6.25 x (157 MB per launch through L2; the 8 x 8 x 64-channel
// tiles of dwconv7_tiled_kernel read 3.06 x), and at 154 KiB of LDS a CU holds one workgroup, whose load, compute and
// reduction phases then run back to back.  Weight prefetch and packed fp32 did not change that (69 us).  Kept as a tested
// operator (tests/test_planes_gpu.py) and as the record of the attempt.
//   * the (TY+6) x (TX+6) x C input halo is staged once in LDS (channel-contiguous: a 16-lane group reads 256 contiguous
//     bytes); the tile shape follows C so that it fits 160 KiB: C <= 96: 8x8, <= 192: 4x8, <= 384: 4x4;
//   * a thread owns a 1 x 4 pixel strip of one 4-channel vector (blockDim = C/4 x TY*TX/4 = 384 for the ConvNeXt widths):
//     per kernel row 10 LDS reads serve 7 taps x 4 outputs; tap-major weights [49][C] come from L2 (75 KiB per stream);
//   * LayerNorm: two-pass (mean, then centred second moment) over per-thread partials through LDS in a FIXED order (no
//     atomics: results are bit-identical run to run), biased variance, per-stream weights.
#include "common.h"

template <int TY, int TX>
__global__ __launch_bounds__(384) void dwconv7_ln_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ lnw, const float* __restrict__ lnb, float eps,
                                                         unsigned short* __restrict__ yp, long ldp, long pstrideB,
                                                         int H, int W, int C, int tilesX, int imgs_per_group) {
  constexpr int HY = TY + 6, HX = TX + 6, NPX = TY * TX, NSTRIP = NPX / 4;
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [HY][HX][C]; reused for the LayerNorm partials
  const int cgN = C >> 2;
  const int b = blockIdx.z;
  if (imgs_per_group > 0) {   // image groups (the two ConvNeXt streams stacked along the batch) with their own weights
    const int grp = b / imgs_per_group;
    w += (long)grp * 49 * C;
    if (bias) bias += (long)grp * C;
    lnw += (long)grp * C;
    lnb += (long)grp * C;
  }
  const int tx0 = (blockIdx.x % tilesX) * TX, ty0 = (blockIdx.x / tilesX) * TY;
  const float* xb = x + (long)b * xstrideB;
  const int nthr = cgN * NSTRIP;   // = blockDim.x
  // ---- halo: every load of a lane is issued before its first LDS write
  {
    const int total = HY * HX * cgN;
    constexpr int MAXIT = 25;   // (HY*HX*cgN) / (cgN*NSTRIP) = HY*HX / NSTRIP: 100/4, 140/8 -> 18, 196/16 -> 13
    float4 v[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = threadIdx.x + it * nthr;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < total) {
        const int pos = i / cgN, cv = i - pos * cgN;
        const int ly = pos / HX, lx = pos - ly * HX;
        const int iy = ty0 + ly - 3, ix = tx0 + lx - 3;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + cv * 4);
      }
    }
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int i = threadIdx.x + it * nthr;
      if (i < total) *reinterpret_cast<float4*>(tile + (long)i * 4) = v[it];
    }
  }
  __syncthreads();
  const int cv = threadIdx.x % cgN, strip = threadIdx.x / cgN;
  const int oy = strip / (TX / 4), ox0 = (strip % (TX / 4)) * 4;
  const int c = cv * 4;
  float4 acc[4];
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = bv;
  // the 7 weight vectors of kernel row kh+1 are fetched (L2) while row kh is computed: with one workgroup per CU there is no
  // other wave to hide a dependent global load behind
  float4 fcur[7], fnext[7];
#pragma unroll
  for (int kw = 0; kw < 7; ++kw) fcur[kw] = *reinterpret_cast<const float4*>(w + (long)kw * C + c);
#pragma unroll 1
  for (int kh = 0; kh < 7; ++kh) {
    const int khn = kh < 6 ? kh + 1 : 6;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) fnext[kw] = *reinterpret_cast<const float4*>(w + (long)(khn * 7 + kw) * C + c);
    float4 in[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) in[i] = *reinterpret_cast<const float4*>(tile + ((long)((oy + kh) * HX + ox0 + i) * cgN + cv) * 4);
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const float4 f = fcur[kw];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        acc[p].x += in[p + kw].x * f.x; acc[p].y += in[p + kw].y * f.y;
        acc[p].z += in[p + kw].z * f.z; acc[p].w += in[p + kw].w * f.w;
      }
    }
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) fcur[kw] = fnext[kw];
  }
  __syncthreads();   // the halo is dead: its memory now carries the LayerNorm partials
  float* part = tile;                 // [NPX][cgN]
  float* tot = tile + NPX * cgN;      // [NPX]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = nthr >> 6;
  float mean[4], rstd[4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float s;
      if (pass == 0) {
        s = (acc[p].x + acc[p].y) + (acc[p].z + acc[p].w);
      } else {
        const float dx = acc[p].x - mean[p], dy = acc[p].y - mean[p], dz = acc[p].z - mean[p], dw = acc[p].w - mean[p];
        s = (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
      part[(strip * 4 + p) * cgN + cv] = s;
    }
    __syncthreads();
    for (int pix = wave; pix < NPX; pix += nwave) {   // fixed summation order: lane-strided partials, then the wave tree
      float s = 0.f;
      for (int j = lane; j < cgN; j += 64) s += part[pix * cgN + j];
      s = wave_sum(s);
      if (lane == 0) tot[pix] = s;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float t = tot[strip * 4 + p] / (float)C;
      if (pass == 0) mean[p] = t;
      else rstd[p] = rsqrtf(t + eps);
    }
    __syncthreads();   // tot / part are rewritten by the second pass
  }
  const int gy = ty0 + oy;
  if (gy >= H) return;
  const float4 gw = *reinterpret_cast<const float4*>(lnw + c), gb = *reinterpret_cast<const float4*>(lnb + c);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int gx = tx0 + ox0 + p;
    if (gx >= W) continue;
    float4 o;
    o.x = (acc[p].x - mean[p]) * rstd[p] * gw.x + gb.x;
    o.y = (acc[p].y - mean[p]) * rstd[p] * gw.y + gb.y;
    o.z = (acc[p].z - mean[p]) * rstd[p] * gw.z + gb.z;
    o.w = (acc[p].w - mean[p]) * rstd[p] * gw.w + gb.w;
    uint2 h2, l2;
    split4(o, h2, l2);
    unsigned short* q_ = yp + (long)b * pstrideB + ((long)gy * W + gx) * ldp + ilv(c);
    *reinterpret_cast<uint2*>(q_) = h2;
    *reinterpret_cast<uint2*>(q_ + 32) = l2;
  }
}

template <int TY, int TX>
static int launch_dwconv7_ln(const float* x, long ldx, long xstrideB, const float* w, const float* bias, const float* lnw,
                             const float* lnb, float eps, unsigned short* yp, long ldp, long pstrideB, int B, int H, int W, int C,
                             int imgs_per_group, hipStream_t stream) {
  const size_t smem = (size_t)(TY + 6) * (TX + 6) * C * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)dwconv7_ln_kernel<TY, TX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int tx = cdiv(W, TX), ty = cdiv(H, TY);
  hipLaunchKernelGGL((dwconv7_ln_kernel<TY, TX>), dim3(tx * ty, 1, B), dim3((C / 4) * (TY * TX / 4)), smem, stream, x, ldx, xstrideB,
                     w, bias, lnw, lnb, eps, yp, ldp, pstrideB, H, W, C, tx, imgs_per_group);
  MMSA_CHECK_LAUNCH("dwconv7_ln");
  return MMSA_OK;
}

extern "C" int mmsa_dwconv7_ln(const float* x, long ldx, long xstrideB, const float* w, const float* bias, const float* lnw,
                               const float* lnb, float eps, unsigned short* yp, long ldp, long pstrideB, int B, int H, int W, int C,
                               int imgs_per_group, hipStream_t stream) {
  MMSA_CHECK_ARG(x && w && lnw && lnb && yp && B > 0 && H > 0 && W > 0, "dwconv7_ln: bad args");
  MMSA_CHECK_ARG(C >= 16 && C <= 384 && (C & 15) == 0, "dwconv7_ln: C must be a multiple of 16 in 16..384 (use dwconv_nhwc + layernorm_rows otherwise), got %d", C);
  MMSA_CHECK_ARG(imgs_per_group >= 0 && (imgs_per_group == 0 || B % imgs_per_group == 0), "dwconv7_ln: bad image grouping");
  MMSA_CHECK_ARG((ldx & 3) == 0 && (xstrideB & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)lnw) | ((uintptr_t)lnb)) & 15) == 0,
                 "dwconv7_ln: operands must be 16-byte aligned");
  MMSA_CHECK_ARG((((uintptr_t)yp) & 127) == 0 && (ldp & 63) == 0 && (pstrideB & 63) == 0 && ldp >= 2L * ((C + 31) / 32 * 32), "dwconv7_ln: bad output planes");
  if (C <= 96) return launch_dwconv7_ln<8, 8>(x, ldx, xstrideB, w, bias, lnw, lnb, eps, yp, ldp, pstrideB, B, H, W, C, imgs_per_group, stream);
  if (C <= 192) return launch_dwconv7_ln<4, 8>(x, ldx, xstrideB, w, bias, lnw, lnb, eps, yp, ldp, pstrideB, B, H, W, C, imgs_per_group, stream);
  return launch_dwconv7_ln<4, 4>(x, ldx, xstrideB, w, bias, lnw, lnb, eps, yp, ldp, pstrideB, B, H, W, C, imgs_per_group, stream);
}
