// Grouped 3x3 convolution (GFE qkv2, AM:88-92: groups = 32, cin_g = cout_g = 9 / 18 / 36 / 72) as an implicit GEMM on the
// fp32 matrix pipe.  Per (image, group, 16 x 16 pixel tile):  D[pixel][co] = sum over (tap, ci) of x[pixel + tap][ci] * w[tap][ci][co],
// i.e. M = 256 pixels, N = cout_g (padded to 16 NT), K = 9 cin_g, walked as 9 taps x (cin_g in chunks of 8 channels = two MFMA
// k-steps of 4).  v_mfma_f32_16x16x4_f32 multiplies and accumulates in fp32, exactly like the FMA kernel it replaces
// (gconv_tiled_kernel<COUT, 3>, one pixel and COUT accumulators per lane, weights through the scalar path: 19 TFLOP/s, 317 us for
// the 1/32-resolution level where 256 workgroups of 4 waves each run alone on their CU).  Measured per launch, ViT-L step:
// cin_g = 72: 317 -> 83 us, 36: 147 -> 116 us, 18: 170 -> 153 us, 9: 276 -> 283 us (stays on the FMA kernel); step -0.33 ms.
//   * A fragment (pixels x 4 channels): lane (l15, kk) reads halo[ci = 4 ks + kk][(row + kh) * 18 + l15 + kw]; channel stride 336
//     floats (= 16 mod 64 banks): the four kk groups of a wave hit four disjoint 16-bank blocks;
//   * B fragment (4 channels x 16 outputs): lane (l15, kk) reads wchunk[tap][4 ks + kk][16 nt + l15]; row stride NTW with
//     NTW = 16 mod 64 (or 48): conflict free; output columns >= cout_g are zero weights;
//   * wave w owns pixel rows 4w .. 4w+3 (4 m-tiles) x NT n-tiles: 4 + NT LDS reads per 4 NT MFMAs.
#include "common.h"

template <int NT>
__global__ __launch_bounds__(256) void gconv3_mfma_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long ldy, int H, int W, int cin_g, int cout_g, int tilesX) {
  constexpr int TW = 18, CCH = 8, CST = 336;                 // halo 18 x 18 = 324 floats per channel, padded to 336
  constexpr int NTW = (NT == 2) ? 48 : NT * 16;              // weight row stride: 16 / 48 / 48 / 80 floats
  __shared__ float halo[CCH * CST];
  __shared__ float wch[9 * CCH * NTW];
  // (dispatch order on purpose: tiles fastest, so XCD k gets tiles k, k + 8, ... of EVERY group, and the groups of a pixel share cache lines --
  // cin_g = 3 ... 9 floats; an XCD-contiguous order, common.h, gave each XCD whole groups and 7 x the reads: profiles/r04_xcd_order.txt)
  const int g = blockIdx.y, b = blockIdx.z;
  const int tx0 = (blockIdx.x % tilesX) * 16, ty0 = (blockIdx.x / tilesX) * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const float* xb = x + (long)b * H * W * ldx + g * cin_g;
  const float* wg = w + (long)g * 9 * cin_g * cout_g;          // [tap][ci][co]
  f32x4 acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int ci0 = 0; ci0 < cin_g; ci0 += CCH) {
    const int nch = min(CCH, cin_g - ci0);
    __syncthreads();   // the previous chunk is fully consumed
    {
      constexpr int NIT = (TW * TW * CCH + 255) / 256;   // 11
      float v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ci = i & (CCH - 1), pos = i >> 3;
        const int ly = pos / TW, lx = pos - ly * TW;
        const int iy = ty0 + ly - 1, ix = tx0 + lx - 1;
        v[it] = 0.f;
        if (pos < TW * TW && ci < nch && iy >= 0 && iy < H && ix >= 0 && ix < W) v[it] = xb[((long)iy * W + ix) * ldx + ci0 + ci];
      }
      // weights of the chunk: [9][CCH][NTW], zero beyond nch channels / cout_g outputs
      constexpr int WN = 9 * CCH * NT * 16;
      constexpr int WIT = (WN + 255) / 256;
      float wv[WIT];
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int co = i % (NT * 16), r = i / (NT * 16);       // r = tap * CCH + ci
        const int ci = r & (CCH - 1), tap = r >> 3;
        wv[it] = 0.f;
        if (i < WN && ci < nch && co < cout_g) wv[it] = wg[((long)tap * cin_g + ci0 + ci) * cout_g + co];
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ci = i & (CCH - 1), pos = i >> 3;
        if (pos < TW * TW) halo[ci * CST + pos] = v[it];
      }
#pragma unroll
      for (int it = 0; it < WIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int co = i % (NT * 16), r = i / (NT * 16);
        if (i < WN) wch[r * NTW + co] = wv[it];
      }
    }
    __syncthreads();
    const int nks = (nch + 3) >> 2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - kh * 3;
      for (int ks = 0; ks < nks; ++ks) {
        const int ci = 4 * ks + kk;
        float bf[NT], af[4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = wch[(tap * CCH + ci) * NTW + nt * 16 + l15];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = halo[ci * CST + (4 * wave + mt + kh) * TW + l15 + kw];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }
  // D: lane (l15 = output column within the n-tile, kk): pixels 4 kk + r of pixel row 4 wave + mt
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int oy = ty0 + 4 * wave + mt;
    if (oy >= H) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = nt * 16 + l15;
      if (co >= cout_g) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ox = tx0 + 4 * kk + r;
        if (ox < W) y[((long)b * H * W + (long)oy * W + ox) * ldy + g * cout_g + co] = acc[mt][nt][r];
      }
    }
  }
}

// internal launcher used by mmsa_gconv_nhwc (conv.hip); returns false when the shape is not covered
bool mmsa_gconv3_mfma_launch(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int G, int cin_g,
                             int cout_g, hipStream_t stream) {
  const int nt = (cout_g + 15) / 16;
  // one n-tile (cout_g <= 16: the 1/4-resolution level, 9 channels per group) is no faster than the FMA kernel (283 vs 276 us:
  // 7 of 16 output columns and 3 of 12 k-slots are padding, and a group's 9 of 288 interleaved channels use 36 B of every line)
  const bool all_nt = MMSA_KNOB("MMSA_GCONV_MFMA_ALL", 0) != 0;
  if (nt < (all_nt ? 1 : 2) || nt > 5 || nt == 4) return false;
  const int tx = cdiv(W, 16), ty = cdiv(H, 16);
  dim3 grid(tx * ty, G, B);
  switch (nt) {
    case 1: hipLaunchKernelGGL((gconv3_mfma_kernel<1>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, cout_g, tx); break;
    case 2: hipLaunchKernelGGL((gconv3_mfma_kernel<2>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, cout_g, tx); break;
    case 3: hipLaunchKernelGGL((gconv3_mfma_kernel<3>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, cout_g, tx); break;
    default: hipLaunchKernelGGL((gconv3_mfma_kernel<5>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, cout_g, tx); break;
  }
  return true;
}
