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
// gram_tn, two kernels and no atomics (round 3).  The first version gave a workgroup one 32 x 32 output tile: X and Y were read c / 32
// times each (3 x at c = 96 ... 24 x at c = 768) and every wave added its 1024 partial sums to G with double-precision atomics --
// 2.4 M atomics per image at level 0, 61 us per launch on average, a memset per call in front.  Now:
//   gram_part_kernel : a workgroup owns 256 rows (as a wave did before: the same fp32 partial sums) and a 96 x 96 block of G (36
//                      accumulator tiles, 9 per wave): X and Y come from HBM c / 96 times; diagonal-head-block mode (nblk > 1) visits only
//                      the blocks a head touches.  The tiles go to a scratch buffer in the MFMA's own lane layout (one float4 per lane
//                      and tile).  (One wave per 96 x 96 block and slice -- 36 tiles, a quarter of the waves -- ran at 100 us per launch:
//                      too few waves to cover the load latency.)
//   gram_sum_kernel  : sums the row slices in DOUBLE, in slice order (16 groups of slices per workgroup, combined through LDS in
//                      fixed order): deterministic by construction, G needs no zeroing.
#define GR_ROWS 256    // rows of X / Y per wave
#define GR_BLK 96      // columns of X / of Y per block
#define GR_T 6         // 16-column tiles per block side
__device__ __forceinline__ bool gram_block_needed(int bi, int bj, int c, int nblk) {
  if (nblk <= 1) return true;
  const int ch = c / nblk;
  const int hi0 = (GR_BLK * bi) / ch, hi1 = min(c - 1, GR_BLK * bi + GR_BLK - 1) / ch;
  const int hj0 = (GR_BLK * bj) / ch, hj1 = min(c - 1, GR_BLK * bj + GR_BLK - 1) / ch;
  return !(hi1 < hj0 || hj1 < hi0);
}
// grid (nb * nb, nslice, B), 256 threads: the four waves of a workgroup share the slice's rows (the second reader of a row hits L1) and
// own 3 x 3 of the block's 6 x 6 accumulator tiles each; part: [B][nb * nb][nslice][36][64] float4
__global__ __launch_bounds__(256) void gram_part_kernel(const float* __restrict__ X, long ldx, const float* __restrict__ Y, long ldy,
                                                        long strideB, float4* __restrict__ part, int P, int c, int nblk, int nslice) {
  const int nb = (c + GR_BLK - 1) / GR_BLK;
  // XCD-contiguous order (common.h): the nb x nb blocks of one row slice read the same rows of X and Y -- they stay on one XCD's L2
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());
  const int bx_ = (int)(wg_ % gridDim.x);
  const int bi = bx_ / nb, bj = bx_ % nb;
  if (!gram_block_needed(bi, bj, c, nblk)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kk = lane >> 4;
  const int b = (int)(wg_ / (gridDim.x * gridDim.y));
  const int slice = (int)((wg_ / gridDim.x) % gridDim.y);
  const int u0 = 3 * (wave >> 1), v0 = 3 * (wave & 1);
  const float* Xb = X + (long)b * strideB;
  const float* Yb = Y + (long)b * strideB;
  // columns beyond c / rows beyond the slice: the load goes to column c - 1 / the slice's first row (always valid) and the value is
  // multiplied by 0 -- UNCONDITIONAL loads, so that 48 of them are in flight (with predicated loads, or a select the compiler may turn
  // back into one, every load sat behind its own exec-masked branch and a full wait: 40 us per launch at every size, pure latency)
  float mi[3], mj[3];
  int ci[3], cj[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int i_ = GR_BLK * bi + 16 * (u0 + u) + l15, j_ = GR_BLK * bj + 16 * (v0 + u) + l15;
    mi[u] = i_ < c ? 1.f : 0.f; mj[u] = j_ < c ? 1.f : 0.f;
    ci[u] = min(i_, c - 1); cj[u] = min(j_, c - 1);
  }
  f32x4 acc[3][3];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) acc[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int pbeg = slice * GR_ROWS;
  const int pend = min(P, pbeg + GR_ROWS);
  // 32 rows per trip: the 48 loads of eight MFMA k-steps are issued before the first MFMA (a `#pragma unroll 8` on the 4-row loop was not
  // honoured: runtime trip count), and the loads of trip i + 1 before the MFMAs of trip i (two register sets)
#define GR_LOADS(av_, bv_, pb_)                                                                   \
  _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                 \
    const int p = (pb_) + 4 * t + kk;                                                             \
    const long pr = p < pend ? p : pbeg;   /* pend - pbeg may not be a multiple of 32 */          \
    const float* xr = Xb + pr * ldx;                                                              \
    const float* yr = Yb + pr * ldy;                                                              \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) { av_[t][u] = xr[ci[u]]; bv_[t][u] = yr[cj[u]]; } \
  }
#define GR_MFMAS(av_, bv_, pb_)                                                                   \
  _Pragma("unroll") for (int t = 0; t < 8; ++t) {                                                 \
    const float mp = ((pb_) + 4 * t + kk) < pend ? 1.f : 0.f;                                     \
    /* value * {1, 0}: a select would let the compiler sink the load back under its condition */  \
    _Pragma("unroll") for (int u = 0; u < 3; ++u) { av_[t][u] *= mp * mi[u]; bv_[t][u] *= mp * mj[u]; } \
    _Pragma("unroll") for (int u = 0; u < 3; ++u)                                                 \
      _Pragma("unroll") for (int v = 0; v < 3; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[t][u], bv_[t][v], acc[u][v], 0, 0, 0); \
  }
  float a0[8][3], b0[8][3], a1[8][3], b1[8][3];
  GR_LOADS(a0, b0, pbeg)
#pragma unroll 1
  for (int pb = pbeg; pb < pend; pb += 64) {   // wave-uniform trip count (MFMA needs full EXEC); rows beyond pend are masked to zero
    __builtin_amdgcn_sched_barrier(0);
    GR_LOADS(a1, b1, pb + 32)
    __builtin_amdgcn_sched_barrier(0);   // loads before the first use (the scheduler otherwise sinks them to their uses: 2-5 in flight)
    GR_MFMAS(a0, b0, pb)
    __builtin_amdgcn_sched_barrier(0);
    GR_LOADS(a0, b0, pb + 64)
    __builtin_amdgcn_sched_barrier(0);
    GR_MFMAS(a1, b1, pb + 32)
  }
#undef GR_LOADS
#undef GR_MFMAS
  float4* dst = part + ((((long)b * nb * nb + bx_) * nslice + slice) * (GR_T * GR_T)) * 64 + lane;
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int v = 0; v < 3; ++v) dst[((u0 + u) * GR_T + v0 + v) * 64] = make_float4(acc[u][v][0], acc[u][v][1], acc[u][v][2], acc[u][v][3]);
}
// grid (36, nb * nb, B), 1024 threads: wave q sums slices q, q + 16, ... of one tile; the 16 sums are then added in order q = 0..15
__global__ __launch_bounds__(1024) void gram_sum_kernel(const float4* __restrict__ part, double* __restrict__ G, int c, int nblk, int nslice) {
  __shared__ double red[16][64][4];
  const int nb = (c + GR_BLK - 1) / GR_BLK;
  const int bi = blockIdx.y / nb, bj = blockIdx.y % nb;
  if (!gram_block_needed(bi, bj, c, nblk)) return;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int tile = blockIdx.x, b = blockIdx.z;
  const float4* src = part + (((long)b * nb * nb + blockIdx.y) * nslice * (GR_T * GR_T) + tile) * 64 + lane;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll 4
  for (int sl = q; sl < nslice; sl += 16) {
    const float4 v = src[(long)sl * (GR_T * GR_T) * 64];
    s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
  }
  red[q][lane][0] = s0; red[q][lane][1] = s1; red[q][lane][2] = s2; red[q][lane][3] = s3;
  __syncthreads();
  if (threadIdx.x < 256) {
    const int ln = threadIdx.x >> 2, r = threadIdx.x & 3;
    double t = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][ln][r];
    const int u = tile / GR_T, v = tile % GR_T;
    const int i = GR_BLK * bi + 16 * u + 4 * (ln >> 4) + r;   // D row = 4 * (lane >> 4) + r
    const int j = GR_BLK * bj + 16 * v + (ln & 15);          // D col = lane & 15
    if (i < c && j < c) G[((long)b * c + i) * c + j] = t;
  }
}

extern "C" long mmsa_gram_tn_scratch_bytes(int B, int P, int c) {
  if (B <= 0 || P <= 0 || c <= 0) return 0;
  const long nb = cdiv(c, GR_BLK);
  return (long)B * nb * nb * cdiv(P, GR_ROWS) * (GR_T * GR_T) * 64 * (long)sizeof(float4);
}

extern "C" int mmsa_gram_tn(const float* X, long ldx, const float* Y, long ldy, long strideB, double* G,
                            int B, int P, int c, int nblk, void* scratch, long scratch_bytes, hipStream_t stream) {
  MMSA_CHECK_ARG(X && Y && G && scratch && B > 0 && P > 0 && c > 0 && nblk > 0 && c % nblk == 0, "gram_tn: bad args");
  MMSA_CHECK_ARG(scratch_bytes >= mmsa_gram_tn_scratch_bytes(B, P, c) && (((uintptr_t)scratch) & 15) == 0,
                 "gram_tn: scratch of %ld bytes, %ld needed (mmsa_gram_tn_scratch_bytes), 16-byte aligned", scratch_bytes, mmsa_gram_tn_scratch_bytes(B, P, c));
  const int nb = cdiv(c, GR_BLK), nslice = cdiv(P, GR_ROWS);
  hipLaunchKernelGGL(gram_part_kernel, dim3(nb * nb, nslice, B), dim3(256), 0, stream, X, ldx, Y, ldy, strideB,
                     reinterpret_cast<float4*>(scratch), P, c, nblk, nslice);
  MMSA_CHECK_LAUNCH("gram_tn");
  hipLaunchKernelGGL(gram_sum_kernel, dim3(GR_T * GR_T, nb * nb, B), dim3(1024), 0, stream, reinterpret_cast<const float4*>(scratch), G, c, nblk, nslice);
  MMSA_CHECK_LAUNCH("gram_tn (slice sum)");
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
  extern __shared__ float attn[];  // [ch][ch+1] logits / softmax, then nq[ch], nk[ch], then this workgroup's weight rows [o_per][ch]
  const int h = blockIdx.x, b = blockIdx.y;
  const int ch = c / heads;
  const int st = ch + 1;
  const double* Gb = G + (long)b * c * c;
  const float t = temp[h];
  // (1) the L2 norms of this head's q and k channels, (2) every logit of the ch x ch block by all 256 threads, (3) the row softmax by
  // one thread per row out of LDS.  Element by element the same expressions as the first version (one thread per ROW doing all three
  // steps: ch dependent double loads per row, 62 us per launch at ch = 96) -- bit-identical results.
  float* nq = attn + ch * st;   // [ch] | nk [ch] behind the logits
  float* nk = nq + ch;
  for (int i = threadIdx.x; i < ch; i += 256) {
    nq[i] = fmaxf((float)sqrt(sq[(long)b * sq_strideB + h * ch + i]), 1e-12f);
    nk[i] = fmaxf((float)sqrt(sk[(long)b * sk_strideB + h * ch + i]), 1e-12f);
  }
  __syncthreads();
  for (int idx0 = threadIdx.x; idx0 < ch * ch; idx0 += 256 * 4) {   // four independent double loads in flight per lane
    double gv[4];
    int ii[4], jj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(idx0 + 256 * u, ch * ch - 1);
      ii[u] = idx / ch; jj[u] = idx - ii[u] * ch;
      gv[u] = Gb[(long)(h * ch + ii[u]) * c + h * ch + jj[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (idx0 + 256 * u < ch * ch) attn[ii[u] * st + jj[u]] = (float)gv[u] / (nq[ii[u]] * nk[jj[u]]) * t;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ch; i += 256) {
    float mx = -INFINITY;
#pragma unroll 8
    for (int j = 0; j < ch; ++j) mx = fmaxf(mx, attn[i * st + j]);
    float s = 0.f;
#pragma unroll 4
    for (int j = 0; j < ch; ++j) {
      const float e = expf(attn[i * st + j] - mx);
      attn[i * st + j] = e;
      s += e;
    }
    const float inv = 1.0f / s;
#pragma unroll 8
    for (int j = 0; j < ch; ++j) attn[i * st + j] *= inv;
  }
  __syncthreads();
  // output rows are split over gridDim.z workgroups (each recomputes the small softmax block above)
  const int o_per = (c + gridDim.z - 1) / gridDim.z;
  const int o_beg = blockIdx.z * o_per, o_end = min(c, o_beg + o_per);
  // this workgroup's rows of the projection weight (columns of head h) go to LDS first: the product loop then runs out of LDS
  // (it read one weight per iteration from global memory: 70 us per launch at ch = 96); the sum over i keeps its order
  float* wl = nk + ch;   // [o_per][ch]
  {
    const int nw = (o_end - o_beg) * ch;
    for (int idx0 = threadIdx.x; idx0 < nw; idx0 += 256 * 4) {   // four independent loads in flight per lane
      float wv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = min(idx0 + 256 * u, nw - 1);
        const int o = idx / ch, i = idx - o * ch;
        wv[u] = Wp[(long)(o_beg + o) * c + h * ch + i];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (idx0 + 256 * u < nw) wl[idx0 + 256 * u] = wv[u];
    }
  }
  __syncthreads();
  for (int idx = o_beg * ch + threadIdx.x; idx < o_end * ch; idx += 256) {
    const int o = idx / ch, j = idx - o * ch;
    const float* wrow = wl + (o - o_beg) * ch;
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < ch; ++i) acc += wrow[i] * attn[i * st + j];   // (unrolled: the LDS reads of eight steps in flight; same sum order)
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
  const int nz = c >= 192 ? 16 : 4;
  const size_t smem = ((size_t)ch * (ch + 1) + 2 * (size_t)ch + (size_t)cdiv(c, nz) * ch) * sizeof(float);   // logits block + the two norm vectors + the workgroup's weight rows
  MMSA_CHECK_ARG(smem <= 64 * 1024, "chanattn_build: head block too large");
  hipLaunchKernelGGL(chanattn_build_kernel, dim3(heads, B, nz), dim3(256), smem, stream, G, sq, sq_strideB, sk, sk_strideB, temp, Wp, planes, c, cpad, heads);
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
// One workgroup per output row (mean over W) or column (mean over H), a lane per 4 channels.  The sums run in the order of the first
// version (element after element: same rounding), but eight rows / columns are LOADED at a time as float4 -- the scalar, one-load-per-
// iteration loop was a chain of up to 256 dependent memory round trips (49-83 us for maps that stream in 5-20).
template <bool VEC>
__global__ __launch_bounds__(256) void pool_hw_kernel(const float* __restrict__ z, long ldz, float* __restrict__ out, long ldo,
                                                      int H, int W, int C) {
  const int r = blockIdx.x, b = blockIdx.y;
  const float* zb = z + (long)b * H * W * ldz;
  const bool row = r < H;
  const int n = row ? W : H;                                   // elements to average
  const long step = row ? ldz : (long)W * ldz;                  // distance between them
  const float* base = zb + (row ? (long)r * W * ldz : (long)(r - H) * ldz);
  constexpr int V = VEC ? 4 : 1;
  for (int c = threadIdx.x * V; c < C; c += 256 * V) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    for (; i + 8 <= n; i += 8) {
      float v[8][4];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if constexpr (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(base + (long)(i + k) * step + c);
          v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
        } else {
          v[k][0] = base[(long)(i + k) * step + c];
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int q = 0; q < V; ++q) s[q] += v[k][q];
    }
    for (; i < n; ++i)
#pragma unroll
      for (int q = 0; q < V; ++q) s[q] += base[(long)i * step + c + q];
    float* o = out + ((long)b * (H + W) + r) * ldo + c;
#pragma unroll
    for (int q = 0; q < V; ++q) o[q] = s[q] / (float)n;
  }
}

extern "C" int mmsa_pool_hw(const float* z, long ldz, float* out, long ldo, int B, int H, int W, int C, hipStream_t stream) {
  MMSA_CHECK_ARG(z && out && B > 0 && H > 0 && W > 0 && C > 0, "pool_hw: bad args");
  if ((C & 3) == 0 && (ldz & 3) == 0 && (ldo & 3) == 0 && ((((uintptr_t)z) | ((uintptr_t)out)) & 15) == 0)
    hipLaunchKernelGGL(pool_hw_kernel<true>, dim3(H + W, B), dim3(256), 0, stream, z, ldz, out, ldo, H, W, C);
  else
    hipLaunchKernelGGL(pool_hw_kernel<false>, dim3(H + W, B), dim3(256), 0, stream, z, ldz, out, ldo, H, W, C);
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
