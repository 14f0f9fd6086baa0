// Normalisation and reduction kernels (HBM-bound; 16-byte vector access, wavefront reductions).
//   layernorm_rows : per-token LayerNorm over channels.  Covers nn.LayerNorm(eps 1e-6) of the ViT blocks
//                    (IE:367,377), injector/extractor norms (AM:479-487,519-520), ConvNeXt LN2d in NHWC
//                    (mmpretrain_custom/models/utils/norm.py:51-90; TC:103-106,329,377) and the
//                    WithBias_LayerNorm of GFE (AM:51-74, eps 1e-5, biased variance).
//   colstats       : per-(batch, channel) sums over the H*W rows of an NHWC map (for GFFM's
//                    nn.LayerNorm(H*W) AM:241,265; F.normalize over HW AM:100-101; avg-pool AM:159).
//   lnhw_apply     : GFFM LayerNorm over the spatial axis + FFRM recalibration, fused apply pass.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// RPW = rows per wave: 1 (64 lanes x NV float4 per row) or 2 (C <= 128: two rows per wave, 32 lanes each -- with one row per
// wave a 96-channel row kept 24 of 64 lanes busy)
template <int NV, int RPW>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(
    const float* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ b, float eps,
    float* __restrict__ y, long ldy, float* __restrict__ y2, long ldy2,
    unsigned short* __restrict__ yp, long ldp, int rows, int C,
    int map_mode, int map_H, int map_W, int group_rows, long w_gstride, long y_gcol, int y_wrap, int plane_fmt, float* __restrict__ clamp_max) {
  constexpr int LPR = 64 / RPW;                       // lanes per row
  float cw_ = 0.f;                                    // clamp watch of the planes output (common.h)
  const int lane = threadIdx.x & (LPR - 1);
  // A wave (or half wave) walks rows slot, slot + nslots, ... with the NEXT row's loads in flight while the current row is
  // reduced and stored: one row per wave made every wave of the launch load, then reduce, then store in lockstep (read burst,
  // then write burst: 2.4 TB/s on a [8192, 1024] map); streamed, loads and stores of different rows overlap.
  const int nslots = gridDim.x * (4 * RPW);
  int row = blockIdx.x * (4 * RPW) + (threadIdx.x / LPR);
  if (row >= rows) return;
  float4 v[NV], vn[NV], wv[NV], bv[NV];
  int cur_grp = -1;
  const bool pair16 = (C & 7) == 0;   // kernel-uniform
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + LPR * i) * 4;
    wv[i] = bv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    vn[i] = c < C ? *reinterpret_cast<const float4*>(x + (long)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (; row < rows; row += nslots) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = vn[i];
    const int nrow = row + nslots;
    if (nrow < rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + LPR * i) * 4;
        if (c < C) vn[i] = *reinterpret_cast<const float4*>(x + (long)nrow * ldx + c);
      }
    }
    // row groups (the two ConvNeXt streams stacked along the rows): group g = row / group_rows has its own weight /
    // bias vectors (w + g * w_gstride) and writes at column offset g * y_gcol; y_wrap: output row = row % group_rows
    const int grp = group_rows > 0 ? row / group_rows : 0;
    if (grp != cur_grp) {   // wave-uniform for RPW == 1; per half wave otherwise (both halves reload: harmless)
      cur_grp = grp;
      const float* wg = w + (long)grp * w_gstride;
      const float* bg = b + (long)grp * w_gstride;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = (lane + LPR * i) * 4;
        if (c < C) { wv[i] = *reinterpret_cast<const float4*>(wg + c); bv[i] = *reinterpret_cast<const float4*>(bg + c); }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);   // lanes beyond C hold zeros
    const float mean = (RPW == 1 ? wave_sum(s) : half_wave_sum(s)) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + LPR * i) * 4;
      if (c < C) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float rstd = 1.0f / sqrtf((RPW == 1 ? wave_sum(q) : half_wave_sum(q)) / (float)C + eps);
    long orow = (group_rows > 0 && y_wrap) ? row - grp * group_rows : row;
    long ocol = (long)grp * y_gcol;
    if (map_mode == 1) {  // 2x2 patchify: token (b,h,w) -> row (b,h/2,w/2), column block (h&1)*2+(w&1)
      const int ww = row % map_W;
      const int t = row / map_W;
      const int hh = t % map_H;
      const int bb = t / map_H;
      orow = ((long)bb * (map_H / 2) + (hh >> 1)) * (map_W / 2) + (ww >> 1);
      ocol += (long)(((hh & 1) << 1) | (ww & 1)) * C;
    }
    float* yr = y ? y + orow * ldy + ocol : nullptr;
    float* y2r = y2 ? y2 + (long)row * ldy2 : nullptr;
    unsigned short* pr = (yp && plane_fmt != MMSA_FMT_H8C) ? yp + orow * ldp : nullptr;   // ilv planes row (ocol is a multiple of 32 when patchifying)
    const bool h8c = yp && plane_fmt == MMSA_FMT_H8C;      // h8c planes: plain [rows, C] output only (checked by the launcher), row pairs
    const H8cRow hr = h8c ? h8c_row(yp, ldp, orow, MMSA_PAD64(C)) : H8cRow{nullptr, nullptr};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = (lane + LPR * i) * 4;
      const bool in = c < C;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (in) {
        o.x = (v[i].x - mean) * rstd * wv[i].x + bv[i].x;
        o.y = (v[i].y - mean) * rstd * wv[i].y + bv[i].y;
        o.z = (v[i].z - mean) * rstd * wv[i].z + bv[i].z;
        o.w = (v[i].w - mean) * rstd * wv[i].w + bv[i].w;
        if (yr) *reinterpret_cast<float4*>(yr + c) = o;
        if (yp) clamp_see(cw_, o);
      }
      if (h8c) {
        if (pair16) h8c_store8_pair<1>(hr, c & ~7, o, lane & 1, in);
        else if (in) h8c_store4(hr, c, o);
      } else if (pr) {
        if (pair16) {
          // C % 8 == 0: lanes 2j / 2j+1 hold channels 8j .. 8j+7: whole-line stores through the lane-pair exchange (common.h)
          store_planes8_pair<1>(pr, (int)ocol + (c & ~7), o, plane_fmt, lane & 1, in);
        } else if (in) {
          store_planes4(pr, (int)ocol + c, o, plane_fmt);
        }
      }
      if (in && y2r) {
        o.x += v[i].x; o.y += v[i].y; o.z += v[i].z; o.w += v[i].w;
        *reinterpret_cast<float4*>(y2r + c) = o;
      }
    }
  }
  if (yp) clamp_report(clamp_max, cw_, mmsa_clamp_limit(plane_fmt));
}

extern "C" int mmsa_layernorm_rows(const float* x, long ldx, const float* w, const float* b, float eps,
                                   float* y, long ldy, float* y2, long ldy2,
                                   unsigned short* yp, long ldp, int rows, int C,
                                   int map_mode, int map_H, int map_W, int group_rows, long w_gstride, long y_gcol, int y_wrap,
                                   int plane_fmt, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(plane_fmt >= MMSA_FMT_B3 && plane_fmt <= MMSA_FMT_F3, "layernorm_rows: bad plane format %d", plane_fmt);
  MMSA_CHECK_ARG(!yp || plane_fmt != MMSA_FMT_H8C || (map_mode == 0 && group_rows == 0 && ldp >= 3L * MMSA_PAD64(C)),
                 "layernorm_rows: h8c planes are a plain [rows, C] output (no patchify / row groups), ldp = pair stride >= 3 * pad64(C)");
  MMSA_CHECK_ARG(x && w && b && (y || yp) && rows > 0 && C > 0, "layernorm_rows: bad args");
  MMSA_CHECK_ARG(!yp || map_mode == 0 || C % 32 == 0, "layernorm_rows: patchified planes need C %% 32 == 0");
  MMSA_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldy2 & 3) == 0 && (ldp & 3) == 0, "layernorm_rows: C/ld must be multiples of 4");
  MMSA_CHECK_ARG((((uintptr_t)yp) & 127) == 0 && (ldp & 63) == 0, "layernorm_rows: planes must be 128-byte aligned, ldp %% 64 == 0");
  MMSA_CHECK_ARG(C <= 4096, "layernorm_rows: C=%d > 4096", C);
  MMSA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w) | ((uintptr_t)b) | ((uintptr_t)y2)) & 15) == 0, "layernorm_rows: pointers must be 16-byte aligned");
  if (map_mode == 1) MMSA_CHECK_ARG(map_H > 0 && map_W > 0 && (map_H & 1) == 0 && (map_W & 1) == 0 && rows % (map_H * map_W) == 0 && y2 == nullptr,
                                    "layernorm_rows: patchify map needs even H,W");
  MMSA_CHECK_ARG(group_rows >= 0 && (group_rows == 0 || (rows % group_rows == 0 && (w_gstride & 3) == 0 && (y_gcol & 3) == 0 && (!yp || (y_gcol & 31) == 0))),
                 "layernorm_rows: bad row grouping");
  MMSA_CHECK_ARG(group_rows == 0 || map_mode == 0 || (group_rows % (map_H * map_W) == 0 && y_wrap == 0 && y_gcol == 0), "layernorm_rows: grouping with patchify needs whole images per group");
  const int rpw = C <= 128 ? 2 : 1;
  // rows per wave slot: 4 once there are enough rows to keep every CU busy that way (streamed: see the kernel); MMSA_LN_ROWS overrides (A/B timing)
  const int ln_rows = MMSA_KNOB("MMSA_LN_ROWS", 0);
  const int per_slot = ln_rows > 0 ? ln_rows : (rows >= 8192 ? 4 : rows >= 4096 ? 2 : 1);
  dim3 grid(cdiv(rows, 4 * rpw * per_slot)), block(256);
#define LN_LAUNCH(NV) hipLaunchKernelGGL((layernorm_rows_kernel<NV, 1>), grid, block, 0, stream, x, ldx, w, b, eps, y, ldy, y2, ldy2, yp, ldp, rows, C, map_mode, map_H, map_W, group_rows, w_gstride, y_gcol, y_wrap, plane_fmt, clamp_max)
  if (C <= 128) hipLaunchKernelGGL((layernorm_rows_kernel<1, 2>), grid, block, 0, stream, x, ldx, w, b, eps, y, ldy, y2, ldy2, yp, ldp, rows, C, map_mode, map_H, map_W, group_rows, w_gstride, y_gcol, y_wrap, plane_fmt, clamp_max);
  else if (C <= 256) LN_LAUNCH(1);
  else if (C <= 512) LN_LAUNCH(2);
  else if (C <= 1024) LN_LAUNCH(4);
  else if (C <= 2048) LN_LAUNCH(8);
  else LN_LAUNCH(16);
#undef LN_LAUNCH
  MMSA_CHECK_LAUNCH("layernorm_rows");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
// colstats: out[b][0][c] = sum_p x, out[b][1][c] = sum_p x^2, out[b][2][c] = sum_p wrow[p]*x   (double)
// grid (ceil(C/64), ceil(HW/ROWS_PER_BLOCK), B), block 256 = 64 channels x 4 row lanes.
#define CS_ROWS 512
// block 256 = 16 float4 columns (64 channels) x 16 row lanes: a wave reads 4 rows x 256 contiguous bytes per instruction
// (the first version read one float per lane: 256 B per wave instruction)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, long ldx, long strideB,
                                                       const float* __restrict__ wrow, int HW, int C,
                                                       double* __restrict__ out) {
  __shared__ double red[3][16][64];
  const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + c4 * 4;
  const int p0 = blockIdx.y * CS_ROWS;
  const int p1 = min(p0 + CS_ROWS, HW);
  const float* xb = x + (long)blockIdx.z * strideB;
  double d1[4] = {0, 0, 0, 0}, d2[4] = {0, 0, 0, 0}, d3[4] = {0, 0, 0, 0};
  if (c < C) {   // C % 4 == 0
    for (int pb = p0 + rg; pb < p1; pb += 16 * 8) {
      float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int p = pb + 16 * k;
        if (p < p1) {
          const float4 v = *reinterpret_cast<const float4*>(xb + (long)p * ldx + c);
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
          if (wrow) { const float wp = wrow[p]; s3.x += wp * v.x; s3.y += wp * v.y; s3.z += wp * v.z; s3.w += wp * v.w; }
        }
      }
      d1[0] += (double)s1.x; d1[1] += (double)s1.y; d1[2] += (double)s1.z; d1[3] += (double)s1.w;
      d2[0] += (double)s2.x; d2[1] += (double)s2.y; d2[2] += (double)s2.z; d2[3] += (double)s2.w;
      d3[0] += (double)s3.x; d3[1] += (double)s3.y; d3[2] += (double)s3.z; d3[3] += (double)s3.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][rg][c4 * 4 + j] = d1[j]; red[1][rg][c4 * 4 + j] = d2[j]; red[2][rg][c4 * 4 + j] = d3[j]; }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
    const int cl = threadIdx.x;
    double a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a1 += red[0][r][cl]; a2 += red[1][r][cl]; a3 += red[2][r][cl]; }
    double* o = out + (long)blockIdx.z * 3 * C;
    const int cc = blockIdx.x * 64 + cl;
    atomicAdd(o + cc, a1);
    atomicAdd(o + C + cc, a2);
    if (wrow) atomicAdd(o + 2 * C + cc, a3);
  }
}

// out_is_zero != 0: the caller has zeroed `out` on this stream already (see mmsa_gram_tn)
extern "C" int mmsa_colstats(const float* x, long ldx, long strideB, const float* wrow, int B, int HW, int C,
                             double* out, int out_is_zero, hipStream_t stream) {
  MMSA_CHECK_ARG(x && out && B > 0 && HW > 0 && C > 0, "colstats: bad args");
  MMSA_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (strideB & 3) == 0 && (((uintptr_t)x) & 15) == 0, "colstats: C / ld must be multiples of 4, x 16-byte aligned");
  if (!out_is_zero && hipMemsetAsync(out, 0, sizeof(double) * 3 * (size_t)B * C, stream) != hipSuccess) {
    mmsa_set_error("colstats: memset failed");
    return MMSA_ERR_LAUNCH;
  }
  dim3 grid(cdiv(C, 64), cdiv(HW, CS_ROWS), B);
  hipLaunchKernelGGL(colstats_kernel, grid, dim3(256), 0, stream, x, ldx, strideB, wrow, HW, C, out);
  MMSA_CHECK_LAUNCH("colstats");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
// FFRM finalize.  From the column stats of the GFFM output F (before its LayerNorm over HW) derive, per channel:
// mean, rstd of the spatial LayerNorm (eps 1e-5, AM:241,265) and the FFRM gate (AM:158-162):
//   avg = mean_p(LN(F))  [analytic: rstd*(sum_p w[p]F/HW - mean*mean(w)) + mean(b)],
//   z = Wc . avg (1x1 conv, no bias), GroupNorm(32) over channels (spatial 1x1), ReLU, sigmoid, mult = 1 + a.
// Three small launches: stats (grid B), matvec (one wave per output channel, grid C/4 x B), gate (grid B).
__global__ __launch_bounds__(256) void ffrm_stats_kernel(const double* __restrict__ stats, int HW, int C, float mean_w, float mean_b,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                         float* __restrict__ avg_o) {
  const int b = blockIdx.x;
  const double* st = stats + (long)b * 3 * C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const double m = st[c] / HW;
    double var = st[C + c] / HW - m * m;
    if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + 1e-5);
    mean_o[(long)b * C + c] = (float)m;
    rstd_o[(long)b * C + c] = (float)rs;
    avg_o[(long)b * C + c] = (float)(rs * (st[2 * C + c] / HW - m * (double)mean_w) + (double)mean_b);
  }
}

__global__ __launch_bounds__(256) void ffrm_matvec_kernel(const float* __restrict__ Wc, const float* __restrict__ avg,
                                                          float* __restrict__ z, int C) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (o >= C) return;
  const float* wr = Wc + (long)o * C;
  const float* av = avg + (long)b * C;
  float s = 0.f;
  for (int i = lane * 4; i < C; i += 256) {
    const float4 w4 = *reinterpret_cast<const float4*>(wr + i);
    const float4 a4 = *reinterpret_cast<const float4*>(av + i);
    s += w4.x * a4.x + w4.y * a4.y + w4.z * a4.z + w4.w * a4.w;
  }
  s = wave_sum(s);
  if (lane == 0) z[(long)b * C + o] = s;
}

__global__ __launch_bounds__(256) void ffrm_gate_kernel(const float* __restrict__ z, const float* __restrict__ gn_w,
                                                        const float* __restrict__ gn_b, float* __restrict__ mult_o, int C) {
  __shared__ float gmean[32], grstd[32];
  const int b = blockIdx.x;
  const float* zb = z + (long)b * C;
  const int cg = C / 32;
  if (threadIdx.x < 32) {
    float m = 0.f;
    for (int i = 0; i < cg; ++i) m += zb[threadIdx.x * cg + i];
    m /= cg;
    float v = 0.f;
    for (int i = 0; i < cg; ++i) {
      const float d = zb[threadIdx.x * cg + i] - m;
      v += d * d;
    }
    gmean[threadIdx.x] = m;
    grstd[threadIdx.x] = 1.0f / sqrtf(v / cg + 1e-5f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gi = c / cg;
    float t = (zb[c] - gmean[gi]) * grstd[gi] * gn_w[c] + gn_b[c];
    t = fmaxf(t, 0.f);
    mult_o[(long)b * C + c] = 1.0f + 1.0f / (1.0f + expf(-t));
  }
}

// scratch: float [2, B, C] (avg, z)
extern "C" int mmsa_ffrm_finalize(const double* stats, int B, int HW, int C, float mean_w, float mean_b,
                                  const float* Wc, const float* gn_w, const float* gn_b,
                                  float* mean_o, float* rstd_o, float* mult_o, float* scratch, hipStream_t stream) {
  MMSA_CHECK_ARG(stats && Wc && gn_w && gn_b && mean_o && rstd_o && mult_o && scratch, "ffrm_finalize: null pointer");
  MMSA_CHECK_ARG(C % 32 == 0 && C <= 8192, "ffrm_finalize: C=%d must be a multiple of 32", C);
  float* avg = scratch;
  float* z = scratch + (long)B * C;
  hipLaunchKernelGGL(ffrm_stats_kernel, dim3(B), dim3(256), 0, stream, stats, HW, C, mean_w, mean_b, mean_o, rstd_o, avg);
  hipLaunchKernelGGL(ffrm_matvec_kernel, dim3(cdiv(C, 4), B), dim3(256), 0, stream, Wc, avg, z, C);
  hipLaunchKernelGGL(ffrm_gate_kernel, dim3(B), dim3(256), 0, stream, z, gn_w, gn_b, mult_o, C);
  MMSA_CHECK_LAUNCH("ffrm_finalize");
  return MMSA_OK;
}

// y[b,p,c] = ((x[b,p,c]-mean[b,c])*rstd[b,c]*w[p] + bias[p]) * mult[b,c]
__global__ __launch_bounds__(256) void lnhw_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ mult,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ y, long ldy, int HW, int C, long total4) {
  const unsigned c4n = C >> 2;   // 32-bit index arithmetic (total4 < 2^32, checked by the launcher)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total4; i += gridDim.x * blockDim.x) {
    const unsigned rowu = i / c4n;
    const int c = (int)(i - rowu * c4n) * 4;
    const long row = rowu;
    const int b = (int)(rowu / (unsigned)HW);
    const int p = (int)(rowu - (unsigned)b * (unsigned)HW);
    const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
    const float4 m = *reinterpret_cast<const float4*>(mean + (long)b * C + c);
    const float4 r = *reinterpret_cast<const float4*>(rstd + (long)b * C + c);
    const float4 mu = *reinterpret_cast<const float4*>(mult + (long)b * C + c);
    const float wp = w[p], bp = bias[p];
    float4 o;
    o.x = ((v.x - m.x) * r.x * wp + bp) * mu.x;
    o.y = ((v.y - m.y) * r.y * wp + bp) * mu.y;
    o.z = ((v.z - m.z) * r.z * wp + bp) * mu.z;
    o.w = ((v.w - m.w) * r.w * wp + bp) * mu.w;
    *reinterpret_cast<float4*>(y + row * ldy + c) = o;
  }
}

extern "C" int mmsa_lnhw_apply(const float* x, long ldx, const float* mean, const float* rstd, const float* mult,
                               const float* w, const float* bias, float* y, long ldy, int B, int HW, int C,
                               hipStream_t stream) {
  MMSA_CHECK_ARG(x && mean && rstd && mult && w && bias && y, "lnhw_apply: null pointer");
  MMSA_CHECK_ARG((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0, "lnhw_apply: C/ld must be multiples of 4");
  const long total4 = (long)B * HW * (C >> 2);
  MMSA_CHECK_ARG(total4 < (1L << 31), "lnhw_apply: too many elements for the 32-bit index arithmetic");
  int blocks = cdiv(total4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(lnhw_apply_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, mean, rstd, mult, w, bias, y, ldy, HW, C, total4);
  MMSA_CHECK_LAUNCH("lnhw_apply");
  return MMSA_OK;
}


// ---- LayerNorm folded into GEMMs (IE:396-421): the producer GEMM of the residual stream leaves, per row and 64-column strip, the sum and
// the sum of squares of what it stored (mmsa_gemm_next_extras); this turns them into (mean, rstd) per row for the consumer's epilogue.
// Strips are combined in double (E[x^2] - mean^2 on fp32 partial sums of 64 values each; biased variance, eps inside the root as nn.LayerNorm).
__global__ void rowstats_finalize_kernel(const float* __restrict__ rs, int rows, int strips, int D, float eps, float* __restrict__ mr) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float2* p = reinterpret_cast<const float2*>(rs) + (long)r * strips;
  double s1 = 0.0, s2 = 0.0;
  for (int i = 0; i < strips; ++i) { const float2 v = p[i]; s1 += (double)v.x; s2 += (double)v.y; }
  const double mean = s1 / D;
  double var = s2 / D - mean * mean;
  var = var < 0.0 ? 0.0 : var;
  reinterpret_cast<float2*>(mr)[r] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

extern "C" int mmsa_rowstats_finalize(const float* rowstats, int rows, int strips, int D, float eps, float* mean_rstd, hipStream_t stream) {
  MMSA_CHECK_ARG(rowstats && mean_rstd && rows > 0 && strips > 0 && D == strips * 64, "rowstats_finalize: bad args (D = 64 * strips)");
  hipLaunchKernelGGL(rowstats_finalize_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, stream, rowstats, rows, strips, D, eps, mean_rstd);
  MMSA_CHECK_LAUNCH("rowstats_finalize");
  return MMSA_OK;
}

// Zero-fill of a scratch range on the caller's stream (the neck's double-precision statistics block): the forward keeps no framework kernel
// inside a captured step.
extern "C" int mmsa_zero_bytes(void* p, size_t bytes, hipStream_t stream) {
  MMSA_CHECK_ARG(p && bytes > 0, "zero_bytes: bad args");
  const hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
  if (e != hipSuccess) {
    mmsa_set_error("zero_bytes: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return MMSA_ERR_LAUNCH;
  }
  return MMSA_OK;
}
