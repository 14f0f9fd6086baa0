// Spatial convolutions of the encoder path on NHWC fp32 maps (stride 1, "same" padding) and the
// NCHW -> patch-matrix gather that turns the strided patchify convolutions into GEMMs.
//   dwconv_nhwc : depthwise k x k (ConvNeXt 7x7 TC:69-70,102; MobileNetV2 dw3x3 AM:288; ConvFFN DWConv AM:459).
//   gconv_nhwc  : grouped conv with cin_g -> cout_g channels per group (GFE qkv1 1x1/g32 AM:87, qkv2 3x3/g32
//                 AM:88, gated-MLP dwconv 3x3 groups=C with 2->2 channels per group AM:123-124).
//   im2col_nchw : patch matrix for PatchEmbed 16x16 s16 (IE:658-663) and the ConvNeXt stem 4x4 s4 (TC:297-304),
//                 K order = (c, kh, kw) = the reference weight's flattened order, zero padded to Kpad.
// These are fp32 VALU kernels (exact fp32 like the reference), 16-byte channel vectors per lane.
#include "common.h"

// weights repacked tap-major: w[(kh*k+kw)*C + c]
__global__ __launch_bounds__(256) void dwconv_nhwc_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, long ldy, long ystrideB,
                                                          unsigned short* __restrict__ yp, long ldp, long pstrideB, int yp_fmt,
                                                          int B, int H, int W, int C, int k, int act, float* __restrict__ clamp_max) {
  // one thread per (pixel, 4 channels); blockIdx.y = image row (b*H + h): 32-bit index arithmetic only
  const int c4n = C >> 2;
  const int pad = k >> 1;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= W * c4n) return;
  {
    const int ww = idx / c4n;
    const int c = (idx - ww * c4n) * 4;
    const int hh = by_ % H, b = by_ / H;
    const float* xb = x + (long)b * xstrideB;
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < k; ++kh) {
      const int ih = hh + kh - pad;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int iw = ww + kw - pad;
        if (iw < 0 || iw >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(xb + ((long)ih * W + iw) * ldx + c);
        const float4 f = *reinterpret_cast<const float4*>(w + (long)(kh * k + kw) * C + c);
        acc.x += v.x * f.x; acc.y += v.y * f.y; acc.z += v.z * f.z; acc.w += v.w * f.w;
      }
    }
    acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
    const long oo = (long)b * ystrideB + ((long)hh * W + ww) * ldy + c;
    if (y) *reinterpret_cast<float4*>(y + oo) = acc;
    if (yp) {   // operand planes, either format (+ clamp watch: common.h)
      float cw_ = 0.f;
      clamp_see(cw_, acc);
      clamp_report(clamp_max, cw_, mmsa_clamp_limit(yp_fmt));
      store_planes4(yp + (long)b * pstrideB + ((long)hh * W + ww) * ldp, c, acc, yp_fmt);
    }
  }
}

// 7x7 depthwise conv (ConvNeXt, TC:69-70,102), LDS-tiled: one workgroup = 8 x 8 output pixels x 64 channels.
// The 14 x 14 x 64 input halo tile is staged once in LDS (50 KiB, channel-contiguous so a 16-lane group reads 256
// contiguous bytes: conflict free); each lane owns a 1 x 4 pixel strip of one 4-channel vector and, per kernel row,
// reads 10 input vectors once for 7 taps x 4 outputs (70 LDS reads instead of 196 global/L1 reads per 4 outputs).
// yp / rs (round 3, the ConvNeXt LayerNorm fold): the conv output also -- or only -- as operand planes (whole-line stores through the
// lane-pair exchange) plus, per pixel and 64-channel chunk, (sum, sum of squares) of the values stored: rs[((img * H * W + pixel) *
// rs_strips + chunk) * 2 + {0, 1}], the strip sums mmsa_rowstats_finalize turns into (mean, rstd) for a row-normalising GEMM epilogue.
// Requires C % 64 == 0, H % 8 == 0, W % 8 == 0 (every lane of the workgroup stays active: checked by the launcher).
// Round 4 measured where its time goes (tools/exp/dw7_abl.hip, profiles/r04_dwconv7_ablation.txt: stage 2, 20.0 us = 12.9 us of loads + stores with one
// tap + 11.2 us of LDS reads / tap weights / FMAs, barely overlapped) and a persistent variant that slides down a column of tiles through a 14-row
// LDS ring with the tap weights in LDS (7 instead of 13 halo loads per lane, next rows requested under the arithmetic): bit-identical, not faster
// (20.3 ... 26 us), so it stayed in tools/exp/.
__global__ __launch_bounds__(256) void dwconv7_tiled_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, long ldy, long ystrideB,
                                                            unsigned short* __restrict__ yp, long ldp, long pstrideB, int yp_fmt,
                                                            float* __restrict__ rs, int rs_strips,
                                                            int H, int W, int C, int tilesX, int imgs_per_group, long w_gstride, float* __restrict__ clamp_max) {
  constexpr int TW = 14, CB = 64;
  // XCD-contiguous tiles (common.h): the tiles of a (channel chunk, image) -- whose halos overlap -- stay on one XCD's L2
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)((wg_ / gridDim.x) % gridDim.y), bz_ = (int)(wg_ / (gridDim.x * gridDim.y));
  if (imgs_per_group > 0) {   // image groups (the two ConvNeXt streams stacked along the batch) with their own weights
    const int grp = bz_ / imgs_per_group;
    w += (long)grp * w_gstride;
    if (bias) bias += (long)grp * C;
  }
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [14][14][64]
  const int b = bz_;
  const int c0 = by_ * CB;
  const int tx0 = (bx_ % tilesX) * 8, ty0 = (bx_ / tilesX) * 8;
  const float* xb = x + (long)b * xstrideB;
  const int cvalid = min(CB, C - c0);   // multiple of 4
  // halo tile: all 13 loads of a lane are issued before the first LDS write (a rolled load -> store loop serialised 13
  // global round trips per lane and was most of the kernel's time)
  {
    constexpr int NIT = (TW * TW * (CB / 4) + 255) / 256;   // 13
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * 256;
      const int cv = i & 15, pos = i >> 4;
      const int ly = pos / TW, lx = pos - ly * TW;
      const int iy = ty0 + ly - 3, ix = tx0 + lx - 3;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < TW * TW * (CB / 4) && cv * 4 < cvalid && iy >= 0 && iy < H && ix >= 0 && ix < W)
        v[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + c0 + cv * 4);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < TW * TW * (CB / 4)) *reinterpret_cast<float4*>(tile + (i >> 4) * CB + (i & 15) * 4) = v[it];
    }
  }
  __syncthreads();
  const int cv = threadIdx.x & 15, strip = threadIdx.x >> 4;   // 16 strips: row = strip>>1, x0 = (strip&1)*4
  const int oy = strip >> 1, ox0 = (strip & 1) * 4;
  const int c = c0 + cv * 4;
  if (cv * 4 >= cvalid) return;
  mmsa_f2 acc01[4], acc23[4];
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < 4; ++p) { acc01[p] = (mmsa_f2){bv.x, bv.y}; acc23[p] = (mmsa_f2){bv.z, bv.w}; }
  // The 49 tap weights come from global memory (L1 / L2 hits), one kernel row = 7 vectors at a time (all 49 would cost occupancy).  Loaded
  // inside the row's loop they were seven dependent load batches per workgroup; now row kh + 1 is requested before row kh is computed
  // (two register sets, the row loop unrolled by two).
  float4 fa[7], fb[7];
#define DW7_LOADW(f_, kh_) _Pragma("unroll") for (int kw = 0; kw < 7; ++kw) f_[kw] = *reinterpret_cast<const float4*>(w + (long)((kh_) * 7 + kw) * C + c);
#define DW7_ROW(f_, kh_)                                                                                                          \
  {                                                                                                                               \
    float4 in[10];                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 10; ++i) in[i] = *reinterpret_cast<const float4*>(tile + ((oy + (kh_)) * TW + ox0 + i) * CB + cv * 4); \
    _Pragma("unroll") for (int kw = 0; kw < 7; ++kw) {                                                                            \
      const mmsa_f2 f01 = {f_[kw].x, f_[kw].y}, f23 = {f_[kw].z, f_[kw].w};                                                       \
      _Pragma("unroll") for (int p = 0; p < 4; ++p) {   /* explicit packed fp32 (v_pk_fma_f32, plain register pairs): the 784 FMAs */ \
        /* of a lane are this kernel's longest stream, 3136 issue cycles per tile as scalar instructions against ~1100 of LDS reads */ \
        const mmsa_f2 i01 = {in[p + kw].x, in[p + kw].y}, i23 = {in[p + kw].z, in[p + kw].w};                                     \
        acc01[p] = __builtin_elementwise_fma(i01, f01, acc01[p]);                                                                 \
        acc23[p] = __builtin_elementwise_fma(i23, f23, acc23[p]);                                                                 \
      }                                                                                                                           \
    }                                                                                                                             \
  }
  DW7_LOADW(fa, 0)
#pragma unroll 1
  for (int kh = 0; kh < 6; kh += 2) {
    __builtin_amdgcn_sched_barrier(0);
    DW7_LOADW(fb, kh + 1)
    __builtin_amdgcn_sched_barrier(0);
    DW7_ROW(fa, kh)
    __builtin_amdgcn_sched_barrier(0);
    DW7_LOADW(fa, kh + 2)
    __builtin_amdgcn_sched_barrier(0);
    DW7_ROW(fb, kh + 1)
  }
  DW7_ROW(fa, 6)
#undef DW7_LOADW
#undef DW7_ROW
  const int gy = ty0 + oy;
  if (gy >= H) return;
  float4 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) acc[p] = make_float4(acc01[p].x, acc01[p].y, acc23[p].x, acc23[p].y);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int gx = tx0 + ox0 + p;
    if (y && gx < W) *reinterpret_cast<float4*>(y + (long)b * ystrideB + ((long)gy * W + gx) * ldy + c) = acc[p];
    if (yp || rs) {   // kernel-uniform; the launcher guarantees full tiles and full 64-channel chunks here
      const long pix = (long)gy * W + gx;
      if (yp) {
        float cw_ = 0.f;   // clamp watch (common.h)
        clamp_see(cw_, acc[p]);
        clamp_report(clamp_max, cw_, mmsa_clamp_limit(yp_fmt));
        store_planes8_pair<1>(yp + (long)b * pstrideB + pix * ldp, c & ~7, acc[p], yp_fmt, (cv & 1) != 0, true);
      }
      if (rs) {     // the 16 lanes cv = 0..15 of this strip hold the pixel's 64 channels of the chunk
        float s1 = (acc[p].x + acc[p].y) + (acc[p].z + acc[p].w);
        float s2 = fmaf(acc[p].x, acc[p].x, acc[p].y * acc[p].y) + fmaf(acc[p].z, acc[p].z, acc[p].w * acc[p].w);
#pragma unroll
        for (int sh = 8; sh > 0; sh >>= 1) { s1 += __shfl_xor(s1, sh, 64); s2 += __shfl_xor(s2, sh, 64); }
        if (cv == 0) *reinterpret_cast<float2*>(rs + (((long)b * H * W + pix) * rs_strips + by_) * 2) = make_float2(s1, s2);
      }
    }
  }
}

// Round 4: the 7 x 7 conv for the plain case (fp32 output only; H, W % 16 == 0, C % 32 == 0: every ConvNeXt stage of the shipped configs).  The tiled
// kernel above spends most of its arithmetic phase waiting for the 49 tap-weight vectors each lane pulls through the vector-memory pipe (64 B/clk per CU:
// 196 KiB per tile against 50 KiB of halo -- profiles/r04_dwconv7_ablation.txt; the 3 x 3 kernels had the same disease).  Here a workgroup owns 16 x 16
// pixels x 32 channels, a lane a 1 x 8 pixel strip of one 4-channel vector: a tap weight is read once for 8 pixels, and from LDS (the chunk's 49 x 32
// weights are staged once per workgroup, 6 KiB); the halo is 22 x 22 (1.9 x the tile instead of 3.06 x).  LDS row pitch 23 pixels: the strips of one
// ds_read_b128 lane group sit in rows r and r + 1, and an odd pitch puts them on different halves of the 64 banks (conflict-free reads).
// Per pixel the same 49 terms in the same order and the same packed FMAs as dwconv7_tiled_kernel: bit-identical.
#define DW7B_PITCH 23
#define DW7B_LDS ((22 * DW7B_PITCH * 32 + 50 * 32) * 4)
// Persistent: gridDim.x workgroups (two per CU: what LDS admits) walk the nt tiles; the NEXT tile's halo and weights are requested into registers before
// the current tile's arithmetic -- which touches LDS only, so nothing in it waits on the vector-memory counter -- and written to LDS after it.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dwconv7_blk_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ y, long ldy, long ystrideB,
                                                          int H, int W, int C, int tilesX, int tilesXY, int nchunk, int nt, int imgs_per_group, long w_gstride) {
  constexpr int TW = 22, CB = 32, PITCH = DW7B_PITCH;
  constexpr int NIT = (TW * TW * (CB / 4) + 255) / 256;   // 16 halo vectors per lane (the last one partly)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                        // [22][PITCH][32]
  float* wl = smem + TW * PITCH * CB;        // [49 taps + the bias][32]
  const int cv = threadIdx.x & 7, blk = threadIdx.x >> 3;   // 32 strips: x0 = 8 * (blk & 1), row = blk >> 1
  const int ox0 = (blk & 1) * 8, oy = blk >> 1;
  float4 v[NIT], wv[2];
  // tile `lin` in dispatch order -> XCD-contiguous work index (common.h) -> (tile in the map, channel chunk, image); issues the tile's loads into v / wv
  int tx0, ty0, c0, bz;
  const float* wg;
  const float* bg;
  auto issue = [&](int lin) {
    const unsigned wk = mmsa_xcd_order((unsigned)lin, (unsigned)nt);
    const int txy = (int)(wk % (unsigned)tilesXY), rest = (int)(wk / (unsigned)tilesXY);
    const int ch = rest % nchunk;
    bz = rest / nchunk;
    c0 = ch * CB;
    tx0 = (txy % tilesX) * 16;
    ty0 = (txy / tilesX) * 16;
    const int grp = imgs_per_group > 0 ? bz / imgs_per_group : 0;   // image groups (the two ConvNeXt streams stacked along the batch) with their own weights
    wg = w + (long)grp * w_gstride;
    bg = bias ? bias + (long)grp * C : nullptr;
    const float* xb = x + (long)bz * xstrideB + c0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * 256;
      const int pos = i >> 3;
      const int ly = pos / TW, lx = pos - ly * TW;
      const int iy = ty0 + ly - 3, ix = tx0 + lx - 3;
      v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < TW * TW * (CB / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W)
        v[it] = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * ldx + (i & 7) * 4);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = threadIdx.x + it * 256;
      wv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < 49 * (CB / 4)) wv[it] = *reinterpret_cast<const float4*>(wg + (long)(i >> 3) * C + c0 + (i & 7) * 4);
      else if (i < 50 * (CB / 4) && bg) wv[it] = *reinterpret_cast<const float4*>(bg + c0 + (i & 7) * 4);   // row 49: the chunk's bias
    }
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = threadIdx.x + it * 256;
      const int pos = i >> 3;
      const int ly = pos / TW, lx = pos - ly * TW;
      if (i < TW * TW * (CB / 4)) *reinterpret_cast<float4*>(tile + (ly * PITCH + lx) * CB + (i & 7) * 4) = v[it];
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = threadIdx.x + it * 256;
      if (i < 50 * (CB / 4)) *reinterpret_cast<float4*>(wl + i * 4) = wv[it];
    }
  };
  int lin = blockIdx.x;
  if (lin >= nt) return;
  issue(lin);
  to_lds();
  __syncthreads();
  for (;;) {
    // this tile's output position and bias (the decode of `issue` is overwritten by the next tile's below)
    float* yo = y + (long)bz * ystrideB + ((long)(ty0 + oy) * W + tx0 + ox0) * ldy + c0 + cv * 4;
    const float4 bv = *reinterpret_cast<const float4*>(wl + 49 * CB + cv * 4);
    const int nxt = lin + gridDim.x;
    const bool more = nxt < nt;
    if (more) issue(nxt);
    mmsa_f2 acc01[8], acc23[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) { acc01[p] = (mmsa_f2){bv.x, bv.y}; acc23[p] = (mmsa_f2){bv.z, bv.w}; }
#pragma unroll 1   // (unrolled it is no faster and, with the 72 prefetch registers live, spills: a scratch reload between the prefetch loads waits for all of them)
    for (int kh = 0; kh < 7; ++kh) {
      float4 in[14], f[7];
      const float* trow = tile + ((oy + kh) * PITCH + ox0) * CB + cv * 4;
#pragma unroll
      for (int i = 0; i < 14; ++i) in[i] = *reinterpret_cast<const float4*>(trow + i * CB);
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) f[kw] = *reinterpret_cast<const float4*>(wl + (kh * 7 + kw) * CB + cv * 4);
#pragma unroll
      for (int kw = 0; kw < 7; ++kw) {
        const mmsa_f2 f01 = {f[kw].x, f[kw].y}, f23 = {f[kw].z, f[kw].w};
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const mmsa_f2 i01 = {in[p + kw].x, in[p + kw].y}, i23 = {in[p + kw].z, in[p + kw].w};
          acc01[p] = __builtin_elementwise_fma(i01, f01, acc01[p]);
          acc23[p] = __builtin_elementwise_fma(i23, f23, acc23[p]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) *reinterpret_cast<float4*>(yo + (long)p * ldy) = make_float4(acc01[p].x, acc01[p].y, acc23[p].x, acc23[p].y);
    if (!more) break;
    lin = nxt;
    __syncthreads();   // every wave is done with this tile's LDS image
    to_lds();
    __syncthreads();
  }
}

// 3 x 3 (the ConvFFN's depthwise conv, AM:446-471): the generic kernel above walks its taps with runtime loops and `continue`s -- two
// dependent loads per tap, nine round trips per thread, 28 us per launch for maps that stream in 7.  Here the nine taps are unrolled,
// every tap's input vector is loaded UNCONDITIONALLY from a clamped position and bit-masked to +0 where the tap lies outside the map
// (fma(+0, f, acc) == acc: the same sum of the same terms in the same order), so the 18 loads of a thread are in flight together.
__global__ __launch_bounds__(256) void dwconv3_nhwc_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, long ldy, long ystrideB,
                                                           unsigned short* __restrict__ yp, long ldp, long pstrideB, int yp_fmt,
                                                           int B, int H, int W, int C, int act, float* __restrict__ clamp_max) {
  const int c4n = C >> 2;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= W * c4n) return;
  const int ww = idx / c4n;
  const int c = (idx - ww * c4n) * 4;
  const int hh = by_ % H, b = by_ / H;
  const float* xb = x + (long)b * xstrideB + c;
  float4 v[9], f[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ih = min(max(hh + t / 3 - 1, 0), H - 1), iw = min(max(ww + t % 3 - 1, 0), W - 1);
    v[t] = *reinterpret_cast<const float4*>(xb + ((long)ih * W + iw) * ldx);
    f[t] = *reinterpret_cast<const float4*>(w + (long)t * C + c);
  }
  __builtin_amdgcn_sched_barrier(0);   // all 18 loads before the first use
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int ih = hh + t / 3 - 1, iw = ww + t % 3 - 1;
    const unsigned m = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? 0xffffffffu : 0u;
    acc.x += __uint_as_float(__float_as_uint(v[t].x) & m) * f[t].x; acc.y += __uint_as_float(__float_as_uint(v[t].y) & m) * f[t].y;
    acc.z += __uint_as_float(__float_as_uint(v[t].z) & m) * f[t].z; acc.w += __uint_as_float(__float_as_uint(v[t].w) & m) * f[t].w;
  }
  acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
  const long oo = (long)b * ystrideB + ((long)hh * W + ww) * ldy + c;
  if (y) *reinterpret_cast<float4*>(y + oo) = acc;
  if (yp) {   // operand planes, either format (+ clamp watch: common.h)
    float cw_ = 0.f;
    clamp_see(cw_, acc);
    clamp_report(clamp_max, cw_, mmsa_clamp_limit(yp_fmt));
    store_planes4(yp + (long)b * pstrideB + ((long)hh * W + ww) * ldp, c, acc, yp_fmt);
  }
}

// The same 3 x 3 conv on a 1 x 4 pixel strip per thread (W % 4 == 0; round 4, as conv_pair.hip's dwpair_gate4_kernel): a kernel row's six input
// vectors are loaded once for the three taps of that row and a tap's weight vector once for the four pixels -- 27 loads for four outputs (6.75 per
// output) instead of 18 per output; every load of a kernel row unconditional (clamped position, zero mask).  Per pixel the same sum of the same nine
// terms in the same order as dwconv3_nhwc_kernel: bit-identical.
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(const float* __restrict__ x, long ldx, long xstrideB,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, long ldy, long ystrideB,
                                                            unsigned short* __restrict__ yp, long ldp, long pstrideB, int yp_fmt,
                                                            int B, int H, int W, int C, int act, float* __restrict__ clamp_max) {
  const int c4n = C >> 2;
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());   // XCD-contiguous rows (common.h)
  const int bx_ = (int)(wg_ % gridDim.x), by_ = (int)(wg_ / gridDim.x);
  const int idx = bx_ * 256 + threadIdx.x;
  if (idx >= (W >> 2) * c4n) return;
  const int wq = idx / c4n;
  const int c = (idx - wq * c4n) * 4;
  const int w0 = wq * 4;
  const int hh = by_ % H, b = by_ / H;
  const float* xb = x + (long)b * xstrideB + c;
  const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[4] = {bv, bv, bv, bv};
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = hh + kh - 1;
    const bool rok = ih >= 0 && ih < H;
    const int ihc = min(max(ih, 0), H - 1);
    float4 v[6], f[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int iw = w0 + i - 1;
      const unsigned m = (rok && iw >= 0 && iw < W) ? 0xffffffffu : 0u;
      const float4 t = *reinterpret_cast<const float4*>(xb + ((long)ihc * W + min(max(iw, 0), W - 1)) * ldx);
      v[i] = make_float4(__uint_as_float(__float_as_uint(t.x) & m), __uint_as_float(__float_as_uint(t.y) & m), __uint_as_float(__float_as_uint(t.z) & m), __uint_as_float(__float_as_uint(t.w) & m));
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) f[kw] = *reinterpret_cast<const float4*>(w + (long)(kh * 3 + kw) * C + c);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        acc[p].x += v[p + kw].x * f[kw].x; acc[p].y += v[p + kw].y * f[kw].y;
        acc[p].z += v[p + kw].z * f[kw].z; acc[p].w += v[p + kw].w * f[kw].w;
      }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float4 o = acc[p];
    o.x = apply_act(o.x, act); o.y = apply_act(o.y, act); o.z = apply_act(o.z, act); o.w = apply_act(o.w, act);
    const long pix = (long)hh * W + w0 + p;
    if (y) *reinterpret_cast<float4*>(y + (long)b * ystrideB + pix * ldy + c) = o;
    if (yp) {   // operand planes, either format (+ clamp watch: common.h)
      float cw_ = 0.f;
      clamp_see(cw_, o);
      clamp_report(clamp_max, cw_, mmsa_clamp_limit(yp_fmt));
      store_planes4(yp + (long)b * pstrideB + pix * ldp, c, o, yp_fmt);
    }
  }
}

extern "C" int mmsa_dwconv_nhwc(const float* x, long ldx, long xstrideB, const float* w, const float* bias,
                                float* y, long ldy, long ystrideB, unsigned short* yp, long ldp, long pstrideB, int yp_fmt,
                                int B, int H, int W, int C, int k, int act, int imgs_per_group, float* rowstats, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(yp_fmt == MMSA_FMT_B3 || yp_fmt == MMSA_FMT_H8 || yp_fmt == MMSA_FMT_F3, "dwconv_nhwc: bad output plane format %d (bf16 hi/lo, h8 lines or f3)", yp_fmt);
  MMSA_CHECK_ARG(x && w && (y || yp) && B > 0 && H > 0 && W > 0 && C > 0, "dwconv_nhwc: bad args");
  MMSA_CHECK_ARG(imgs_per_group >= 0 && (imgs_per_group == 0 || B % imgs_per_group == 0), "dwconv_nhwc: bad image grouping");
  MMSA_CHECK_ARG(!yp || ((((uintptr_t)yp) & 127) == 0 && (ldp & 63) == 0 && (pstrideB & 63) == 0), "dwconv_nhwc: bad output planes");
  MMSA_CHECK_ARG((k & 1) == 1 && k <= 7, "dwconv_nhwc: odd kernel <= 7 expected, got %d", k);
  MMSA_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && (xstrideB & 3) == 0 && (ystrideB & 3) == 0, "dwconv_nhwc: C/ld must be multiples of 4");
  const bool extras = yp || rowstats;   // planes / strip-sum outputs exist on the tiled 7 x 7 kernel only, on full tiles and chunks
  MMSA_CHECK_ARG(!rowstats || (k == 7 && act == ACT_NONE), "dwconv_nhwc: rowstats are written by the 7x7 kernel without activation only");
  if (k == 7 && (y || yp) && act == ACT_NONE && ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w) | ((uintptr_t)bias)) & 15) == 0 &&
      (!extras || ((C & 63) == 0 && (H & 7) == 0 && (W & 7) == 0))) {
    if (y && !extras && (C & 31) == 0 && (H & 15) == 0 && (W & 15) == 0 && MMSA_KNOB("MMSA_DWCONV7_BLK", 1) != 0) {
      static MmsaPerDevice per_dev_ = {};
      const int num_cus = mmsa_per_device(per_dev_, [] { (void)hipFuncSetAttribute((const void*)dwconv7_blk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DW7B_LDS); });
      const int txy = (W >> 4) * (H >> 4), nchunk = C >> 5;
      const long nt = (long)txy * nchunk * B;
      MMSA_CHECK_ARG(nt < (1L << 30), "dwconv_nhwc: too many tiles");
      const int grid = (int)(nt < 2L * num_cus ? nt : 2L * num_cus);   // two resident workgroups per CU (71 KiB of LDS each)
      hipLaunchKernelGGL(dwconv7_blk_kernel, dim3(grid), dim3(256), DW7B_LDS, stream, x, ldx, xstrideB, w, bias, y, ldy, ystrideB, H, W, C, W >> 4, txy, nchunk,
                         (int)nt, imgs_per_group, (long)k * k * C);
      MMSA_CHECK_LAUNCH("dwconv_nhwc(7x7 blocks)");
      return MMSA_OK;
    }
    const int tx = cdiv(W, 8), ty = cdiv(H, 8);
    dim3 grid(tx * ty, cdiv(C, 64), B);
    hipLaunchKernelGGL(dwconv7_tiled_kernel, grid, dim3(256), 14 * 14 * 64 * sizeof(float), stream, x, ldx, xstrideB, w, bias, y, ldy, ystrideB,
                       yp, ldp, pstrideB, yp_fmt, rowstats, C >> 6, H, W, C, tx, imgs_per_group, (long)k * k * C, clamp_max);
    MMSA_CHECK_LAUNCH("dwconv_nhwc(7x7 tiled)");
    return MMSA_OK;
  }
  MMSA_CHECK_ARG(!rowstats, "dwconv_nhwc: rowstats need the tiled 7x7 kernel (C %% 64 == 0, H, W %% 8 == 0, 16-byte aligned pointers)");
  MMSA_CHECK_ARG(imgs_per_group == 0 || imgs_per_group == B, "dwconv_nhwc: image groups are implemented by the tiled 7x7 kernel only");
  MMSA_CHECK_ARG((long)B * H <= 65535, "dwconv_nhwc: B*H too large for the launch grid");
  const bool generic3 = MMSA_KNOB("MMSA_DWCONV3_GENERIC", 0) != 0;   // A/B aid
  if (k == 3 && !generic3 && (W & 3) == 0 && MMSA_KNOB("MMSA_DWCONV3_STRIP", 1) != 0 && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)y)) & 15) == 0)
    hipLaunchKernelGGL(dwconv3_strip_kernel, dim3(cdiv((long)(W >> 2) * (C >> 2), 256), B * H), dim3(256), 0, stream, x, ldx, xstrideB, w, bias, y, ldy, ystrideB, yp, ldp, pstrideB, yp_fmt, B, H, W, C, act, clamp_max);
  else if (k == 3 && !generic3 && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)bias) | ((uintptr_t)y)) & 15) == 0)
    hipLaunchKernelGGL(dwconv3_nhwc_kernel, dim3(cdiv((long)W * (C >> 2), 256), B * H), dim3(256), 0, stream, x, ldx, xstrideB, w, bias, y, ldy, ystrideB, yp, ldp, pstrideB, yp_fmt, B, H, W, C, act, clamp_max);
  else
    hipLaunchKernelGGL(dwconv_nhwc_kernel, dim3(cdiv((long)W * (C >> 2), 256), B * H), dim3(256), 0, stream, x, ldx, xstrideB, w, bias, y, ldy, ystrideB, yp, ldp, pstrideB, yp_fmt, B, H, W, C, k, act, clamp_max);
  MMSA_CHECK_LAUNCH("dwconv_nhwc");
  return MMSA_OK;
}

// Grouped conv.  Weights repacked as w[g][tap][ci][co] (co fastest).  One lane per (pixel, output channel);
// lanes of a wavefront cover consecutive output channels -> coalesced weight reads and stores, broadcast-ish
// activation reads (all output channels of a group read the same cin_g inputs).
__global__ __launch_bounds__(256) void gconv_nhwc_kernel(const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, long ldy,
                                                         int B, int H, int W, int G, int cin_g, int cout_g, int k, int act) {
  const int Cout = G * cout_g;
  const long total = (long)B * H * W * Cout;
  const int pad = k >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int oc = (int)(i % Cout);
    long t = i / Cout;
    const int ww = (int)(t % W);
    t /= W;
    const int hh = (int)(t % H);
    const int b = (int)(t / H);
    const int g = oc / cout_g, co = oc - g * cout_g;
    const float* wg = w + (long)g * k * k * cin_g * cout_g + co;
    float acc = bias ? bias[oc] : 0.f;
    for (int kh = 0; kh < k; ++kh) {
      const int ih = hh + kh - pad;
      if (ih < 0 || ih >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        const int iw = ww + kw - pad;
        if (iw < 0 || iw >= W) continue;
        const float* xp = x + (((long)b * H + ih) * W + iw) * ldx + g * cin_g;
        const float* wp = wg + (long)(kh * k + kw) * cin_g * cout_g;
        for (int ci = 0; ci < cin_g; ++ci) acc += xp[ci] * wp[(long)ci * cout_g];
      }
    }
    y[(((long)b * H + hh) * W + ww) * ldy + oc] = apply_act(acc, act);
  }
}

// Tiled grouped conv for the GFE qkv convs (groups = 32, cin_g / cout_g = 3..72): one workgroup = a 16x16 pixel
// tile of one (image, group).  The group's input halo tile is staged through LDS in chunks of 8 input channels
// (layout [ci][row][col]: a wavefront reads consecutive pixels -> conflict free), every lane keeps COUT
// accumulators, and the weights -- identical for all lanes -- come through the scalar path (wave-uniform index),
// so the inner loop is one LDS read per COUT FMAs.
template <int COUT, int K>
__global__ __launch_bounds__(256) void gconv_tiled_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          float* __restrict__ y, long ldy, int H, int W, int cin_g, int tilesX) {
  constexpr int PAD = K / 2, TW = 16 + 2 * PAD, CCH = 8;
  __shared__ float tile[CCH][TW * TW];
  // (dispatch order on purpose: tiles fastest, so XCD k gets tiles k, k + 8, ... of EVERY group, and the groups of a pixel share cache lines --
  // cin_g = 3 ... 9 floats; an XCD-contiguous order, common.h, gave each XCD whole groups and 7 x the reads: profiles/r04_xcd_order.txt)
  const int g = blockIdx.y, b = blockIdx.z;
  const int tx0 = (blockIdx.x % tilesX) * 16, ty0 = (blockIdx.x / tilesX) * 16;
  const int px = threadIdx.x & 15, py = threadIdx.x >> 4;
  const float* xb = x + (long)b * H * W * ldx + g * cin_g;
  const float* wg = w + (long)g * K * K * cin_g * COUT;
  float acc[COUT];
#pragma unroll
  for (int i = 0; i < COUT; ++i) acc[i] = 0.f;
  for (int ci0 = 0; ci0 < cin_g; ci0 += CCH) {
    const int nch = min(CCH, cin_g - ci0);
    __syncthreads();
    {   // all loads of a lane in flight before the first LDS write (a rolled load -> store loop serialised ~11 global
        // round trips per chunk and was most of the kernel's time)
      constexpr int NIT = (TW * TW * CCH + 255) / 256;
      float v[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ci = i & (CCH - 1), pos = i >> 3;
        const int ly = pos / TW, lx = pos - ly * TW;
        const int iy = ty0 + ly - PAD, ix = tx0 + lx - PAD;
        v[it] = 0.f;
        if (pos < TW * TW && ci < nch && iy >= 0 && iy < H && ix >= 0 && ix < W) v[it] = xb[((long)iy * W + ix) * ldx + ci0 + ci];
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int i = threadIdx.x + it * 256;
        const int ci = i & (CCH - 1), pos = i >> 3;
        if (pos < TW * TW) tile[ci][pos] = v[it];
      }
    }
    __syncthreads();
    for (int ci = 0; ci < nch; ++ci) {
#pragma unroll
      for (int kh = 0; kh < K; ++kh)
#pragma unroll
        for (int kw = 0; kw < K; ++kw) {
          const float xv = tile[ci][(py + kh) * TW + px + kw];
          const float* wp = wg + ((long)(kh * K + kw) * cin_g + ci0 + ci) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] += xv * wp[co];
        }
    }
  }
  const int oy = ty0 + py, ox = tx0 + px;
  if (oy < H && ox < W) {
    float* yp = y + ((long)b * H * W + (long)oy * W + ox) * ldy + g * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) yp[co] = acc[co];
  }
}

// The pair-conv kernels (dwpair_nhwc_kernel, dwpair_gate_kernel) live in conv_pair.hip: that file is compiled without the SLP
// vectoriser (build.py).
int mmsa_dwpair_nhwc_launch(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int C, hipStream_t stream);

bool mmsa_gconv3_mfma_launch(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int G, int cin_g,
                             int cout_g, hipStream_t stream);   // gconv_mfma.hip

template <int COUT>
static void launch_gconv_tiled(const float* x, long ldx, const float* w, float* y, long ldy, int B, int H, int W, int G,
                               int cin_g, int k, hipStream_t stream) {
  const int tx = cdiv(W, 16), ty = cdiv(H, 16);
  dim3 grid(tx * ty, G, B);
  if (k == 1)
    hipLaunchKernelGGL((gconv_tiled_kernel<COUT, 1>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, tx);
  else
    hipLaunchKernelGGL((gconv_tiled_kernel<COUT, 3>), grid, dim3(256), 0, stream, x, ldx, w, y, ldy, H, W, cin_g, tx);
}

extern "C" int mmsa_gconv_nhwc(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy,
                               int B, int H, int W, int G, int cin_g, int cout_g, int k, int act, hipStream_t stream) {
  MMSA_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && G > 0 && cin_g > 0 && cout_g > 0, "gconv_nhwc: bad args");
  MMSA_CHECK_ARG((k & 1) == 1 && k <= 7, "gconv_nhwc: odd kernel <= 7 expected, got %d", k);
  const bool no_mfma = MMSA_KNOB("MMSA_GCONV_VALU", 0) != 0;   // A/B aid: the FMA kernels for the 3x3 case too
  if (!bias && act == ACT_NONE && k == 3 && !no_mfma && mmsa_gconv3_mfma_launch(x, ldx, w, y, ldy, B, H, W, G, cin_g, cout_g, stream)) {
    MMSA_CHECK_LAUNCH("gconv_nhwc(3x3 mfma)");
    return MMSA_OK;
  }
  if (!bias && act == ACT_NONE && (k == 1 || k == 3)) {
    bool done = true;
    switch (cout_g) {
      case 3: launch_gconv_tiled<3>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 6: launch_gconv_tiled<6>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 9: launch_gconv_tiled<9>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 12: launch_gconv_tiled<12>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 18: launch_gconv_tiled<18>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 24: launch_gconv_tiled<24>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 36: launch_gconv_tiled<36>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      case 72: launch_gconv_tiled<72>(x, ldx, w, y, ldy, B, H, W, G, cin_g, k, stream); break;
      default: done = false;
    }
    if (done) {
      MMSA_CHECK_LAUNCH("gconv_nhwc(tiled)");
      return MMSA_OK;
    }
    if (k == 3 && cin_g == 2 && cout_g == 2 && (G & 1) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 &&
        ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w)) & 15) == 0) {
      MMSA_CHECK_ARG((long)B * H <= 65535, "gconv_nhwc(pair): B*H too large for the launch grid");
      return mmsa_dwpair_nhwc_launch(x, ldx, w, y, ldy, B, H, W, G * 2, stream);
    }
  }
  const long total = (long)B * H * W * G * cout_g;
  int blocks = cdiv(total, 256);
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL(gconv_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, w, bias, y, ldy, B, H, W, G, cin_g, cout_g, k, act);
  MMSA_CHECK_LAUNCH("gconv_nhwc");
  return MMSA_OK;
}

// out[(b,ph,pw)][(c,kh,kw)] = x[b, c0 + c, ph*p + kh, pw*p + kw]; columns >= Cin*p*p are zero.
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, int Ctot, int c0, int Cin,
                                                          int H, int W, int p, float* __restrict__ out, int Kpad, long total) {
  const int Hp = H / p, Wp = W / p;
  const int Kreal = Cin * p * p;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kpad);
    long t = i / Kpad;
    float v = 0.f;
    if (kk < Kreal) {
      const int pw = (int)(t % Wp);
      const long t2 = t / Wp;
      const int ph = (int)(t2 % Hp);
      const int b = (int)(t2 / Hp);
      const int kw = kk % p;
      const int kh = (kk / p) % p;
      const int c = kk / (p * p);
      v = x[(((long)b * Ctot + c0 + c) * H + ph * p + kh) * W + pw * p + kw];
    }
    out[i] = v;
  }
}

extern "C" int mmsa_im2col_nchw(const float* x, int B, int Ctot, int c0, int Cin, int H, int W, int p,
                                float* out, int Kpad, hipStream_t stream) {
  MMSA_CHECK_ARG(x && out && B > 0 && Cin > 0 && c0 >= 0 && c0 + Cin <= Ctot && p > 0, "im2col_nchw: bad args");
  MMSA_CHECK_ARG(H % p == 0 && W % p == 0 && Kpad >= Cin * p * p, "im2col_nchw: H,W must be multiples of the patch; Kpad >= Cin*p*p");
  const long total = (long)B * (H / p) * (W / p) * Kpad;
  int blocks = cdiv(total, 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(im2col_nchw_kernel, dim3(blocks), dim3(256), 0, stream, x, Ctot, c0, Cin, H, W, p, out, Kpad, total);
  MMSA_CHECK_LAUNCH("im2col_nchw");
  return MMSA_OK;
}
