// Kernels specific to the RoadFormer2Neck modality fusion (AM:297-394) on NHWC fp32 maps.
//   gram_tn        : G[b] = X[b]^T Y[b] over the H*W rows, exact fp32 on v_mfma_f32_16x16x4_f32 per 256-row slice; the slices
//                    are combined in DOUBLE (atomics): their order is arbitrary, and with fp32 atomics the 1e-7 noise of the
//                    sum was amplified by GFFM's softmax over energies of magnitude ~1e4 to run-to-run differences of up to
//                    2e-3 in c1 / c2 (GFE channel attention q k^T AM:102; GFFM energies AM:252-253).
//   chanattn_build : GFE: L2-normalised, temperature-scaled channel softmax per head (AM:100-103) folded with
//                    the 1x1 `proj` (AM:107) into one per-image [c,c] weight, emitted as bf16 hi/lo planes so
//                    that  proj(attn @ v)  becomes a single split3 GEMM over the tokens.
//   gffm_build     : GFFM: row softmax of E and of E^T (AM:254-255) as bf16 hi/lo planes.
//   gelu_gate      : gated-MLP  gelu(x1) * x2  (AM:129-130).
//   pool_hw / ca_apply : CoordinateAttention pooling and gating (AM:187-201, 218-221).
#include "common.h"
#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
#define GR_ROWS 1024   // rows of X/Y per workgroup (4 waves x 256)
__global__ __launch_bounds__(256) void gram_tn_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                                      long strideB, double* __restrict__ G, int P, int c, int nblk) {
  const int nt = (c + 31) / 32;
  const int ti = blockIdx.x / nt, tj = blockIdx.x % nt;
  if (nblk > 1) {  // only tiles that intersect a diagonal head block are needed
    const int ch = c / nblk;
    const int hi0 = (32 * ti) / ch, hi1 = min(c - 1, 32 * ti + 31) / ch;
    const int hj0 = (32 * tj) / ch, hj1 = min(c - 1, 32 * tj + 31) / ch;
    if (hi1 < hj0 || hj1 < hi0) return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const int b = blockIdx.z;
  const float* Xb = X + (long)b * strideB;
  const float* Yb = Y + (long)b * strideB;
  const int i0 = 32 * ti + l15, i1 = i0 + 16;
  const int j0 = 32 * tj + l15, j1 = j0 + 16;
  const bool vi0 = i0 < c, vi1 = i1 < c, vj0 = j0 < c, vj1 = j1 < c;
  f32x4 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int pbeg = blockIdx.y * GR_ROWS + wave * (GR_ROWS / 4);
  const int pend = min(P, pbeg + GR_ROWS / 4);
  // unrolled x8 so that 32 independent loads are in flight per lane (rolled, every iteration waited for its own 4 loads)
#pragma unroll 8
  for (int pb = pbeg; pb < pend; pb += 4) {   // wave-uniform trip count (MFMA needs full EXEC)
    const int p = pb + kk;
    const bool vp = p < pend;  // pend - pbeg may not be a multiple of 4
    const float* xr = Xb + (long)(vp ? p : pbeg) * ldx;
    const float* yr = Yb + (long)(vp ? p : pbeg) * ldy;
    const float a0 = (vp && vi0) ? xr[i0] : 0.f;
    const float a1 = (vp && vi1) ? xr[i1] : 0.f;
    const float b0 = (vp && vj0) ? yr[j0] : 0.f;
    const float b1 = (vp && vj1) ? yr[j1] : 0.f;
    acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
  }
  double* Gb = G + (long)b * c * c;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 32 * ti + 16 * u + 4 * kk + r;   // D row = 4*(lane>>4) + r
        const int j = 32 * tj + 16 * v + l15;          // D col = lane & 15
        if (i < c && j < c) atomicAdd(Gb + (long)i * c + j, (double)acc[u][v][r]);
      }
}

// out_is_zero != 0: the caller has zeroed G on this stream already (backbone: ONE memset per neck level covers the six accumulator
// buffers of the level, instead of one memset node per call: 24 -> 4 per forward)
extern "C" int mmsa_gram_tn(const float* X, long ldx, const float* Y, long ldy, long strideB, double* G,
                            int B, int P, int c, int nblk, int out_is_zero, hipStream_t stream) {
  MMSA_CHECK_ARG(X && Y && G && B > 0 && P > 0 && c > 0 && nblk > 0 && c % nblk == 0, "gram_tn: bad args");
  if (!out_is_zero && hipMemsetAsync(G, 0, sizeof(double) * (size_t)B * c * c, stream) != hipSuccess) {
    mmsa_set_error("gram_tn: memset failed");
    return MMSA_ERR_LAUNCH;
  }
  const int nt = cdiv(c, 32);
  dim3 grid(nt * nt, cdiv(P, GR_ROWS), B);
  hipLaunchKernelGGL(gram_tn_kernel, grid, dim3(256), 0, stream, X, ldx, Y, ldy, strideB, G, P, c, nblk);
  MMSA_CHECK_LAUNCH("gram_tn");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
// grid (heads, B).  G: [B,c,c] = q^T k; sq/sk: column sums of squares of q and k (double, from colstats slot 1).
// planes: [B, c, cpad] bf16 hi/lo of  Wcomb[o][j] = sum_i Wp[o][i] * attn[i][j]   (attn block diagonal per head).
__global__ __launch_bounds__(256) void chanattn_build_kernel(const double* __restrict__ G, const double* __restrict__ sq,
                                                             long sq_strideB, const double* __restrict__ sk, long sk_strideB,
                                                             const float* __restrict__ temp, const float* __restrict__ Wp,
                                                             unsigned short* __restrict__ planes,
                                                             int c, int cpad, int heads) {
  extern __shared__ float attn[];  // [ch][ch+1]
  const int h = blockIdx.x, b = blockIdx.y;
  const int ch = c / heads;
  const int st = ch + 1;
  const double* Gb = G + (long)b * c * c;
  const float t = temp[h];
  for (int i = threadIdx.x; i < ch; i += 256) {
    const int gi = h * ch + i;
    const float nq = fmaxf((float)sqrt(sq[(long)b * sq_strideB + gi]), 1e-12f);
    float mx = -INFINITY;
    for (int j = 0; j < ch; ++j) {
      const int gj = h * ch + j;
      const float nk = fmaxf((float)sqrt(sk[(long)b * sk_strideB + gj]), 1e-12f);
      const float v = (float)Gb[(long)gi * c + gj] / (nq * nk) * t;
      attn[i * st + j] = v;
      mx = fmaxf(mx, v);
    }
    float s = 0.f;
    for (int j = 0; j < ch; ++j) {
      const float e = expf(attn[i * st + j] - mx);
      attn[i * st + j] = e;
      s += e;
    }
    const float inv = 1.0f / s;
    for (int j = 0; j < ch; ++j) attn[i * st + j] *= inv;
  }
  __syncthreads();
  // output rows are split over gridDim.z workgroups (each recomputes the small softmax block above)
  const int o_per = (c + gridDim.z - 1) / gridDim.z;
  const int o_beg = blockIdx.z * o_per, o_end = min(c, o_beg + o_per);
  for (int idx = o_beg * ch + threadIdx.x; idx < o_end * ch; idx += 256) {
    const int o = idx / ch, j = idx - o * ch;
    const float* wrow = Wp + (long)o * c + h * ch;
    float acc = 0.f;
    for (int i = 0; i < ch; ++i) acc += wrow[i] * attn[i * st + j];
    unsigned short hh, ll;
    split_bf16(acc, hh, ll);
    unsigned short* q_ = planes + ((long)b * c + o) * 2 * cpad + ilv(h * ch + j);   // ilv planes [B, c, 2*cpad]
    q_[0] = hh;
    q_[32] = ll;
  }
}

extern "C" int mmsa_chanattn_build(const double* G, const double* sq, long sq_strideB, const double* sk, long sk_strideB,
                                   const float* temp, const float* Wp, unsigned short* planes,
                                   int B, int c, int cpad, int heads, hipStream_t stream) {
  MMSA_CHECK_ARG(G && sq && sk && temp && Wp && planes, "chanattn_build: null pointer");
  MMSA_CHECK_ARG(c % heads == 0 && cpad >= c, "chanattn_build: bad channel split");
  const int ch = c / heads;
  const size_t smem = (size_t)ch * (ch + 1) * sizeof(float);
  MMSA_CHECK_ARG(smem <= 64 * 1024, "chanattn_build: head block too large");
  hipLaunchKernelGGL(chanattn_build_kernel, dim3(heads, B, c >= 192 ? 16 : 4), dim3(256), smem, stream, G, sq, sq_strideB, sk, sk_strideB, temp, Wp, planes, c, cpad, heads);
  MMSA_CHECK_LAUNCH("chanattn_build");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
// grid (c, B, 2): z = 0 -> Ax[i][:] = softmax_j E[i][j];  z = 1 -> Ay[i][:] = softmax_j E[j][i]
__global__ __launch_bounds__(256) void gffm_build_kernel(const double* __restrict__ E, unsigned short* __restrict__ xp,
                                                         unsigned short* __restrict__ yp, int c, int cpad) {
  __shared__ float red[4];
  const int i = blockIdx.x, b = blockIdx.y, tr = blockIdx.z;
  const double* Eb = E + (long)b * c * c;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the row maximum as fp32 (exact comparison: fp32 values are a subset), the shifted logits from the double energies
  float mx = -INFINITY;
  for (int j = threadIdx.x; j < c; j += 256) mx = fmaxf(mx, (float)(tr ? Eb[(long)j * c + i] : Eb[(long)i * c + j]));
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int j = threadIdx.x; j < c; j += 256) s += expf((float)((tr ? Eb[(long)j * c + i] : Eb[(long)i * c + j]) - (double)mx));
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  unsigned short* pl = tr ? yp : xp;   // ilv planes [B, c, 2*cpad]
  for (int j = threadIdx.x; j < c; j += 256) {
    const float p = expf((float)((tr ? Eb[(long)j * c + i] : Eb[(long)i * c + j]) - (double)mx)) * inv;
    unsigned short hh, ll;
    split_bf16(p, hh, ll);
    unsigned short* q_ = pl + ((long)b * c + i) * 2 * cpad + ilv(j);
    q_[0] = hh;
    q_[32] = ll;
  }
}

extern "C" int mmsa_gffm_build(const double* E, unsigned short* xp, unsigned short* yp, int B, int c, int cpad,
                               hipStream_t stream) {
  MMSA_CHECK_ARG(E && xp && yp && cpad >= c, "gffm_build: bad args");
  hipLaunchKernelGGL(gffm_build_kernel, dim3(c, B, 2), dim3(256), 0, stream, E, xp, yp, c, cpad);
  MMSA_CHECK_LAUNCH("gffm_build");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_gate_kernel(const float* __restrict__ x, long ldx, float* __restrict__ y, long ldy,
                                                        int C, long total4) {
  const unsigned c4n = C >> 2;   // 32-bit index arithmetic (total4 < 2^32, checked by the launcher): 64-bit div/mod cost more than the op
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total4; i += gridDim.x * blockDim.x) {
    const unsigned rowu = i / c4n;
    const int c = (int)(i - rowu * c4n) * 4;
    const long row = rowu;
    const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c);
    const float4 g = *reinterpret_cast<const float4*>(x + row * ldx + C + c);
    float4 o;
    o.x = apply_act(a.x, ACT_GELU) * g.x;
    o.y = apply_act(a.y, ACT_GELU) * g.y;
    o.z = apply_act(a.z, ACT_GELU) * g.z;
    o.w = apply_act(a.w, ACT_GELU) * g.w;
    *reinterpret_cast<float4*>(y + row * ldy + c) = o;
  }
}

extern "C" int mmsa_gelu_gate(const float* x, long ldx, float* y, long ldy, long rows, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(x && y && rows > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0, "gelu_gate: bad args");
  const long total4 = rows * (C >> 2);
  MMSA_CHECK_ARG(total4 < (1L << 31), "gelu_gate: too many elements for the 32-bit index arithmetic");
  int blocks = cdiv(total4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gelu_gate_kernel, dim3(blocks), dim3(256), 0, stream, x, ldx, y, ldy, C, total4);
  MMSA_CHECK_LAUNCH("gelu_gate");
  return MMSA_OK;
}

// ---------------------------------------------------------------------------------------------
// out[b][r][c], r < H: mean_w z[b,r,w,c];  r >= H: mean_h z[b,h,r-H,c]     (cat of pool_h and pool_w, AM:190-192)
__global__ __launch_bounds__(256) void pool_hw_kernel(const float* __restrict__ z, long ldz, float* __restrict__ out, long ldo,
                                                      int H, int W, int C) {
  const int r = blockIdx.x, b = blockIdx.y;
  const float* zb = z + (long)b * H * W * ldz;
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    if (r < H) {
      for (int w = 0; w < W; ++w) s += zb[((long)r * W + w) * ldz + c];
      s /= (float)W;
    } else {
      const int w = r - H;
      for (int h = 0; h < H; ++h) s += zb[((long)h * W + w) * ldz + c];
      s /= (float)H;
    }
    out[((long)b * (H + W) + r) * ldo + c] = s;
  }
}

extern "C" int mmsa_pool_hw(const float* z, long ldz, float* out, long ldo, int B, int H, int W, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(z && out && B > 0 && H > 0 && W > 0 && C > 0, "pool_hw: bad args");
  hipLaunchKernelGGL(pool_hw_kernel, dim3(H + W, B), dim3(256), 0, stream, z, ldz, out, ldo, H, W, C);
  MMSA_CHECK_LAUNCH("pool_hw");
  return MMSA_OK;
}

// out = z + z * a_w[b,w,c] * a_h[b,h,c];  att: [B, H+W, C] rows 0..H-1 = a_h, H.. = a_w
__global__ __launch_bounds__(256) void ca_apply_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ att, long lda,
                                                       float* __restrict__ out, long ldo, unsigned short* __restrict__ outp, long ldp,
                                                       int H, int W, int C, long total4) {
  const unsigned c4n = C >> 2;   // 32-bit index arithmetic (total4 < 2^32, checked by the launcher)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < (unsigned)total4; i += gridDim.x * blockDim.x) {
    const unsigned rowu = i / c4n;
    const int c = (int)(i - rowu * c4n) * 4;
    const long row = rowu;
    const unsigned t = rowu / (unsigned)W;
    const int w = (int)(rowu - t * (unsigned)W);
    const int b = (int)(t / (unsigned)H);
    const int h = (int)(t - (unsigned)b * (unsigned)H);
    const float4 v = *reinterpret_cast<const float4*>(z + row * ldz + c);
    const float4 ah = *reinterpret_cast<const float4*>(att + ((long)b * (H + W) + h) * lda + c);
    const float4 aw = *reinterpret_cast<const float4*>(att + ((long)b * (H + W) + H + w) * lda + c);
    float4 o;
    o.x = v.x + v.x * aw.x * ah.x;
    o.y = v.y + v.y * aw.y * ah.y;
    o.z = v.z + v.z * aw.z * ah.z;
    o.w = v.w + v.w * aw.w * ah.w;
    if (out) *reinterpret_cast<float4*>(out + row * ldo + c) = o;
    if (outp) {
      uint2 h2, l2;
      split4(o, h2, l2);
      unsigned short* q_ = outp + row * ldp + ilv(c);
      *reinterpret_cast<uint2*>(q_) = h2;
      *reinterpret_cast<uint2*>(q_ + 32) = l2;
    }
  }
}

extern "C" int mmsa_ca_apply(const float* z, long ldz, const float* att, long lda, float* out, long ldo,
                             unsigned short* out_planes, long ldp, int B, int H, int W, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(z && att && (out || out_planes) && (C & 3) == 0 && (ldz & 3) == 0 && (lda & 3) == 0 && (ldo & 3) == 0, "ca_apply: bad args");
  MMSA_CHECK_ARG(!out_planes || ((((uintptr_t)out_planes) & 127) == 0 && (ldp & 63) == 0 && ldp >= 2L * ((C + 31) / 32 * 32)), "ca_apply: bad output planes");
  const long total4 = (long)B * H * W * (C >> 2);
  MMSA_CHECK_ARG(total4 < (1L << 31), "ca_apply: too many elements for the 32-bit index arithmetic");
  int blocks = cdiv(total4, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(ca_apply_kernel, dim3(blocks), dim3(256), 0, stream, z, ldz, att, lda, out, ldo, out_planes, ldp, H, W, C, total4);
  MMSA_CHECK_LAUNCH("ca_apply");
  return MMSA_OK;
}
