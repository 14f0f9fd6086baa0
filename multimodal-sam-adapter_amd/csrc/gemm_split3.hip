// split3 GEMM for gfx950:  C[M,N] = epilogue( A[M,K] x W[N,K]^T ),  W as bf16 hi/lo planes,
// A either fp32 (split to hi/lo while staged) or already bf16 hi/lo planes written by the producing kernel.
//
// Every Linear / 1x1-conv / patchify-conv / conv-transpose of the encoder path goes through this
// kernel (reference call sites: IE:488,499,162-167 qkv/proj/MLP; TC:107-111 pointwise convs;
// AM:947-950 fc1-4; OPS/modules/ms_deform_attn.py:103-129 value/offset/output projections;
// BK:55,324 `up`).  The reference computes these in fp32; single-pass bf16 misses the 1e-3 parity
// gate, so each operand is split x = hi + lo (two bf16, ~16 mantissa bits) and three
// MFMA products hi*hi + hi*lo + lo*hi are accumulated in fp32.
//
// Tiling: 128x128 block tile, BK = 32 (= one v_mfma_f32_16x16x32_bf16 k-step), 4 waves (2x2), each
// wave 64x64 = 4x4 MFMA tiles x 3 products.  The MFMA "A" operand is the WEIGHT tile and the "B"
// operand the ACTIVATION tile, so a lane's 4 accumulator registers are 4 consecutive output columns
// -> 16-byte epilogue loads/stores.  LDS image per plane: [k-chunk g=0..3][row 0..127][8 bf16], which
// makes every ds_read_b128 fragment read bank-conflict free (16 lanes x 16 B = one 256-B bank row).
// Two LDS stages + register prefetch: one barrier per k-tile.
//
// Activation planes (A_PLANES): an hi/lo bf16 pair costs the same 4 bytes per element as fp32 but is split ONCE
// by the producer (LayerNorm, GELU epilogue, attention, MSDA, depthwise conv) instead of once per column block
// of every consumer GEMM, and is staged with plain 16-byte copies (no VALU work in the main loop).
// The epilogue can emit fp32, planes, or both.
#include "common.h"

struct GemmArgs {
  const float* A; const unsigned short* Ap; long lda; long strideA;   // Ap: ilv planes (lda in bf16 units)
  const unsigned short* Wp; long strideW;                            // ilv planes, row stride 2*K
  const float* bias; long strideBias;
  const float* colscale;
  const float* resid; long ldr; long strideR; int resid_mod; float beta;
  float* C; long ldc; long strideC;
  unsigned short* Cp; long ldcp; long strideCp;                        // ilv planes output
  int M, N, K;
  int act; float alpha;
  int out_mode; int ps_H, ps_W, ps_C;   // out_mode 1: 2x2 pixel-shuffle store (conv-transpose 2x2 s2)
  int cp_fmt;                           // format of the planes output (common.h: MMSA_FMT_B3 / MMSA_FMT_H8)
  float* clamp_max;                     // optional clamp watch word (common.h)
};

#define BM 128
#define BN 128
#define BK 32
#define PLANE_BYTES (4 * 128 * 16)     // 8 KiB per bf16 plane per stage
#define STAGE_BYTES (4 * PLANE_BYTES)  // Ahi, Alo, Whi, Wlo

// GEN = true compiles in the rarely used index arithmetic (pixel-shuffle store, broadcast residual): integer
// divisions per output element that the common epilogue must not pay for.
// F16: the 16-bit hi/lo pairs are fp16 halves ("f3", common.h) -- A is split that way while staged, the weight planes come that way, fp16 MFMAs
template <bool AP, bool GEN, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_split3_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  const int bz = blockIdx.z;
  const int K = a.K;

  // ---- global load assignment
  // fp32 activations: 4 x float4 per thread: row = (tid>>3) + 32*i, float col = (tid&7)*4
  // plane operands (weights always, activations when AP): 2 x uint4 per plane per thread:
  //   row = (tid&7) + 8*(tid>>5) + 64*i, k-chunk = (tid>>3)&3 -- each 8-lane ds_write_b128 group then writes
  //   128 contiguous LDS bytes (8 rows of one chunk): conflict free.  (row = tid>>2, chunk = tid&3 would put
  //   4 lanes of every group on the same banks: chunk stride 2048 B = 0 mod 128.)
  const int p_row = (tid & 7) + 8 * (tid >> 5), p_chunk = (tid >> 3) & 3;
  // fp32 activations: lane -> (half = tid&1, row = ((tid>>1)&7) + 8*(tid>>6), chunk = (tid>>4)&3): each 16-lane
  // ds_write_b64 group writes 128 contiguous bytes
  const int f_row = ((tid >> 1) & 7) + 8 * (tid >> 6), f_chunk = (tid >> 4) & 3, f_half = tid & 1;
  const float* a_ptr0 = nullptr; const float* a_ptr1 = nullptr; const float* a_ptr2 = nullptr; const float* a_ptr3 = nullptr;
  const unsigned short* ah_ptr0 = nullptr; const unsigned short* ah_ptr1 = nullptr;
  const unsigned short* al_ptr0 = nullptr; const unsigned short* al_ptr1 = nullptr;
  if constexpr (AP) {
    const unsigned short* Ap = a.Ap + (long)bz * a.strideA;
    int r0 = m0 + p_row, r1 = r0 + 64;
    r0 = r0 < a.M ? r0 : a.M - 1;
    r1 = r1 < a.M ? r1 : a.M - 1;
    ah_ptr0 = Ap + (long)r0 * a.lda + p_chunk * 8; ah_ptr1 = Ap + (long)r1 * a.lda + p_chunk * 8;
    al_ptr0 = ah_ptr0 + 32; al_ptr1 = ah_ptr1 + 32;
  } else {
    const float* A = a.A + (long)bz * a.strideA;
    const int a_c4 = f_chunk * 8 + f_half * 4;
    int r0 = m0 + f_row, r1 = r0 + 32, r2 = r0 + 64, r3 = r0 + 96;
    r0 = r0 < a.M ? r0 : a.M - 1; r1 = r1 < a.M ? r1 : a.M - 1;
    r2 = r2 < a.M ? r2 : a.M - 1; r3 = r3 < a.M ? r3 : a.M - 1;
    a_ptr0 = A + (long)r0 * a.lda + a_c4; a_ptr1 = A + (long)r1 * a.lda + a_c4;
    a_ptr2 = A + (long)r2 * a.lda + a_c4; a_ptr3 = A + (long)r3 * a.lda + a_c4;
  }
  const int a_lds_off = f_chunk * 2048 + f_row * 16 + f_half * 8;  // fp32 path, + 512*i
  const unsigned short* Wp = a.Wp + (long)bz * a.strideW;
  int wr0 = n0 + p_row, wr1 = wr0 + 64;
  wr0 = wr0 < a.N ? wr0 : a.N - 1;
  wr1 = wr1 < a.N ? wr1 : a.N - 1;
  const unsigned short* w_hi_ptr0 = Wp + (long)wr0 * 2 * K + p_chunk * 8;
  const unsigned short* w_hi_ptr1 = Wp + (long)wr1 * 2 * K + p_chunk * 8;
  const unsigned short* w_lo_ptr0 = w_hi_ptr0 + 32;
  const unsigned short* w_lo_ptr1 = w_hi_ptr1 + 32;
  const int p_lds_off = p_chunk * 2048 + p_row * 16;  // plane operands, + 1024*i

  float4 ra0, ra1, ra2, ra3;            // fp32 activations in flight
  uint4 rah0, rah1, ral0, ral1;         // plane activations in flight
  uint4 rwh0, rwh1, rwl0, rwl1;         // weights in flight

#define LOAD_GLOBAL(k0)                                                   \
  do {                                                                    \
    if constexpr (AP) {                                                   \
      rah0 = *reinterpret_cast<const uint4*>(ah_ptr0 + 2 * (k0));         \
      rah1 = *reinterpret_cast<const uint4*>(ah_ptr1 + 2 * (k0));         \
      ral0 = *reinterpret_cast<const uint4*>(al_ptr0 + 2 * (k0));         \
      ral1 = *reinterpret_cast<const uint4*>(al_ptr1 + 2 * (k0));         \
    } else {                                                              \
      ra0 = *reinterpret_cast<const float4*>(a_ptr0 + (k0));              \
      ra1 = *reinterpret_cast<const float4*>(a_ptr1 + (k0));              \
      ra2 = *reinterpret_cast<const float4*>(a_ptr2 + (k0));              \
      ra3 = *reinterpret_cast<const float4*>(a_ptr3 + (k0));              \
    }                                                                     \
    rwh0 = *reinterpret_cast<const uint4*>(w_hi_ptr0 + 2 * (k0));         \
    rwh1 = *reinterpret_cast<const uint4*>(w_hi_ptr1 + 2 * (k0));         \
    rwl0 = *reinterpret_cast<const uint4*>(w_lo_ptr0 + 2 * (k0));         \
    rwl1 = *reinterpret_cast<const uint4*>(w_lo_ptr1 + 2 * (k0));         \
  } while (0)

#define STORE_A(base, reg, i)                                                               \
  do {                                                                                      \
    uint2 hi_, lo_;                                                                         \
    if constexpr (F16) f3_split4(reg, hi_, lo_); else split4(reg, hi_, lo_);               \
    *reinterpret_cast<uint2*>((base) + 0 * PLANE_BYTES + a_lds_off + (i) * 512) = hi_;      \
    *reinterpret_cast<uint2*>((base) + 1 * PLANE_BYTES + a_lds_off + (i) * 512) = lo_;      \
  } while (0)

#define STORE_LDS(stage_)                                                              \
  do {                                                                                 \
    unsigned char* sb_ = smem + (stage_) * STAGE_BYTES;                                \
    if constexpr (AP) {                                                                \
      *reinterpret_cast<uint4*>(sb_ + 0 * PLANE_BYTES + p_lds_off) = rah0;             \
      *reinterpret_cast<uint4*>(sb_ + 0 * PLANE_BYTES + p_lds_off + 1024) = rah1;      \
      *reinterpret_cast<uint4*>(sb_ + 1 * PLANE_BYTES + p_lds_off) = ral0;             \
      *reinterpret_cast<uint4*>(sb_ + 1 * PLANE_BYTES + p_lds_off + 1024) = ral1;      \
    } else {                                                                           \
      STORE_A(sb_, ra0, 0); STORE_A(sb_, ra1, 1); STORE_A(sb_, ra2, 2); STORE_A(sb_, ra3, 3); \
    }                                                                                  \
    *reinterpret_cast<uint4*>(sb_ + 2 * PLANE_BYTES + p_lds_off) = rwh0;               \
    *reinterpret_cast<uint4*>(sb_ + 2 * PLANE_BYTES + p_lds_off + 1024) = rwh1;        \
    *reinterpret_cast<uint4*>(sb_ + 3 * PLANE_BYTES + p_lds_off) = rwl0;               \
    *reinterpret_cast<uint4*>(sb_ + 3 * PLANE_BYTES + p_lds_off + 1024) = rwl1;        \
  } while (0)

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int frag_a_off = g * 2048 + (wm * 64 + l15) * 16;  // + mi*256
  const int frag_w_off = g * 2048 + (wn * 64 + l15) * 16;  // + ni*256

  const int nk = K / BK;
  LOAD_GLOBAL(0);
  STORE_LDS(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nk) LOAD_GLOBAL((kt + 1) * BK);
    const unsigned char* base = smem + stage * STAGE_BYTES;
    bf16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(base + 0 * PLANE_BYTES + frag_a_off + i * 256);
      al[i] = *reinterpret_cast<const bf16x8*>(base + 1 * PLANE_BYTES + frag_a_off + i * 256);
      wh[i] = *reinterpret_cast<const bf16x8*>(base + 2 * PLANE_BYTES + frag_w_off + i * 256);
      wl[i] = *reinterpret_cast<const bf16x8*>(base + 3 * PLANE_BYTES + frag_w_off + i * 256);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        if constexpr (F16) {
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[ni]), __builtin_bit_cast(f16x8, ah[mi]), acc[ni][mi], 0, 0, 0);
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ni]), __builtin_bit_cast(f16x8, al[mi]), acc[ni][mi], 0, 0, 0);
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ni]), __builtin_bit_cast(f16x8, ah[mi]), acc[ni][mi], 0, 0, 0);
        } else {
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);
        }
      }
    if (kt + 1 < nk) STORE_LDS(stage ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+l15][n = ..+4g .. +3]
  const float* bias = a.bias ? a.bias + (long)bz * a.strideBias : nullptr;
  const float* colscale = a.colscale ? a.colscale + (long)bz * a.strideBias : nullptr;   // per-column vectors share the batch stride
  const float* resid = a.resid ? a.resid + (long)bz * a.strideR : nullptr;
  float* C = a.C ? a.C + (long)bz * a.strideC : nullptr;
  unsigned short* Cp = a.Cp ? a.Cp + (long)bz * a.strideCp : nullptr;
  const bool vec_ok = ((a.ldc & 3) == 0) && (!resid || (a.ldr & 3) == 0) && ((a.N & 3) == 0) && ((a.ldcp & 3) == 0);
  // column parameters of this lane's 4x4 columns, loaded once, unconditionally (clamped index): the element loop
  // must not contain loads (each would cost a dependent s_waitcnt vmcnt(0))
  float bv[4][4], cv[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int nn = min(n0 + wn * 64 + ni * 16 + 4 * g + r, a.N - 1);
      int ci = nn;
      if constexpr (GEN) { if (a.out_mode == 1) ci %= a.ps_C; }
      bv[ni][r] = bias ? bias[nn] : 0.f;
      cv[ni][r] = colscale ? colscale[ci] * a.alpha : a.alpha;
    }
  // ROLLED over the four 16-row sub-tiles (the code always takes accumulator column 0, the columns are then rotated down
  // by register moves), one activation branch per 4 values: fully unrolled with the activation switch expanded per
  // element this epilogue was > 100 KiB of straight-line code -- beyond the 64 KiB instruction cache -- and ran at
  // instruction-fetch speed (same finding as gemm_v2.hip).
  float cw_ = 0.f;   // clamp watch of the planes output (common.h)
#pragma unroll 1
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + l15;
    f32x4 col[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      col[ni] = acc[ni][0];
      acc[ni][0] = acc[ni][1];
      acc[ni][1] = acc[ni][2];
      acc[ni][2] = acc[ni][3];
    }
    if (m >= a.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + 4 * g;
      if (n >= a.N) continue;
      float v[4] = {col[ni][0], col[ni][1], col[ni][2], col[ni][3]};
      long drow = m;
      int dcol = n;
      long rrow = m;
      if constexpr (GEN) {
        if (a.out_mode == 1) {
          const int ij = n / a.ps_C;
          dcol = n - ij * a.ps_C;
          const int w_ = m % a.ps_W;
          const int t_ = m / a.ps_W;
          const int h_ = t_ % a.ps_H;
          const int b_ = t_ / a.ps_H;
          drow = ((long)(b_ * 2 * a.ps_H + 2 * h_ + (ij >> 1))) * (2 * a.ps_W) + 2 * w_ + (ij & 1);
        }
        rrow = a.resid_mod > 0 ? (long)((int)drow % a.resid_mod) : drow;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += bv[ni][r];
      if (a.act != ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= cv[ni][r];
      if (vec_ok && n + 3 < a.N) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (resid) {
          const float4 rr = *reinterpret_cast<const float4*>(resid + rrow * a.ldr + dcol);
          o.x += a.beta * rr.x; o.y += a.beta * rr.y; o.z += a.beta * rr.z; o.w += a.beta * rr.w;
        }
        if (C) *reinterpret_cast<float4*>(C + drow * a.ldc + dcol) = o;
        if (Cp) { clamp_see(cw_, o); store_planes4(Cp + drow * a.ldcp, dcol, o, MMSA_CP_AT(a.cp_fmt, dcol)); }
      } else {
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          if (n + r < a.N) {
            float x = v[r];
            if (resid) x += a.beta * resid[rrow * a.ldr + dcol + r];
            if (C) C[drow * a.ldc + dcol + r] = x;
            if (Cp) { clamp_see1(cw_, x); store_planes1(Cp + drow * a.ldcp, dcol + r, x, MMSA_CP_AT(a.cp_fmt, dcol + r)); }
          }
        }
      }
    }
  }
  if (Cp) clamp_report(a.clamp_max, cw_, MMSA_CP_SPLIT(a.cp_fmt) ? MMSA_H8_MAX : mmsa_clamp_limit(MMSA_CP_BASE(a.cp_fmt)));
}

int mmsa_gemm_v2_launch(const unsigned short* Ap, long lda, long strideA,
                        const unsigned short* Wp, long strideW,
                        const float* bias, long strideBias, const float* colscale,
                        const float* resid, long ldr, long strideR, int resid_mod, float beta,
                        float* C, long ldc, long strideC,
                        unsigned short* Cp, long ldcp, long strideCp,
                        int M, int N, int K, int batch, int act, float alpha,
                        int out_mode, int ps_H, int ps_W, int ps_C, int fmt, int cp_fmt, int max_grid, hipStream_t stream,
                        float* rs_out, const float* rn_mr, const float* rn_cs, int flavour, float* clamp_max);

// ---- tiny problems (CoordinateAttention's 1x1 convs on pooled maps, AM:187-201: M = B*(h+w) <= ~1000 rows, N or K of 8..48): a
// 128 x 128 MFMA tile would be one or two workgroups walking K alone (64 us for M = 128, N = 48, K = 1536).  Here one wave owns
// (row m, 8 output columns): lanes stride K with float4 loads of A and 8-byte loads of the weight planes (w = hi + lo, exact),
// plain fp32 FMAs, a xor-shuffle reduction and the usual epilogue.  Same contract as the big kernels (fp32 A, fp32 C).
__global__ __launch_bounds__(256) void gemm_tiny_kernel(GemmArgs a, int ncg) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long per_b = (long)a.M * ncg;
  const int bz = blockIdx.y;
  if (wv >= per_b) return;
  const int m = (int)(wv / ncg), n0 = (int)(wv - (long)m * ncg) * 8;
  const float* A = a.A + (long)bz * a.strideA + (long)m * a.lda;
  const unsigned short* Wp = a.Wp + (long)bz * a.strideW;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int k = lane * 4; k < a.K; k += 256) {
    const float4 x = *reinterpret_cast<const float4*>(A + k);
    const int ko = ilv(k);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = min(n0 + j, a.N - 1);
      const unsigned short* wr = Wp + (long)n * 2 * a.K + ko;
      const uint2 h = *reinterpret_cast<const uint2*>(wr), l = *reinterpret_cast<const uint2*>(wr + 32);
      const float w0 = __uint_as_float(h.x << 16) + __uint_as_float(l.x << 16);
      const float w1 = __uint_as_float(h.x & 0xFFFF0000u) + __uint_as_float(l.x & 0xFFFF0000u);
      const float w2 = __uint_as_float(h.y << 16) + __uint_as_float(l.y << 16);
      const float w3 = __uint_as_float(h.y & 0xFFFF0000u) + __uint_as_float(l.y & 0xFFFF0000u);
      acc[j] = fmaf(x.x, w0, fmaf(x.y, w1, fmaf(x.z, w2, fmaf(x.w, w3, acc[j]))));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = wave_sum(acc[j]);
  if (lane < 8 && n0 + lane < a.N) {
    const int n = n0 + lane;
    float v = acc[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) v = lane == j ? acc[j] : v;
    if (a.bias) v += a.bias[(long)bz * a.strideBias + n];
    v = apply_act(v, a.act) * (a.colscale ? a.colscale[(long)bz * a.strideBias + n] * a.alpha : a.alpha);
    if (a.resid) v += a.beta * a.resid[(long)bz * a.strideR + (long)m * a.ldr + n];
    a.C[(long)bz * a.strideC + (long)m * a.ldc + n] = v;
  }
}


// C-ABI entry: see include/mmsa.h for the contract.
extern "C" int mmsa_gemm_split3(const float* A, const unsigned short* Ap, long lda, long strideA,
                                const unsigned short* Wp, long strideW,
                                const float* bias, long strideBias, const float* colscale,
                                const float* resid, long ldr, long strideR, int resid_mod, float beta,
                                float* C, long ldc, long strideC,
                                unsigned short* Cp, long ldcp, long strideCp,
                                int M, int N, int K, int batch, int act, float alpha,
                                int out_mode, int ps_H, int ps_W, int ps_C, int fmt, int cp_fmt, int max_grid,
                                float* rowstats_out, const float* rownorm_mean_rstd, const float* rownorm_colsum, int flavour,
                                float* clamp_max, hipStream_t stream) {
  const bool ap = Ap != nullptr;
  float* const rs_out = rowstats_out;
  const float* const rn_mr = rownorm_mean_rstd;
  const float* const rn_cs = rownorm_colsum;
  MMSA_CHECK_ARG(!rn_mr == !rn_cs, "gemm_split3: mean/rstd rows and column sums go together");
  MMSA_CHECK_ARG(!(rs_out && rn_mr), "gemm_split3: a GEMM either writes row statistics or normalises by them");
  const bool extras = rs_out || rn_mr;
  MMSA_CHECK_ARG(!extras || (ap && M >= 128), "gemm_split3: row statistics / row normalisation need activation planes and M >= 128");
  MMSA_CHECK_ARG(fmt >= MMSA_FMT_B3 && fmt <= MMSA_FMT_F3 && cp_fmt >= 0 && MMSA_CP_BASE(cp_fmt) >= MMSA_FMT_B3 && MMSA_CP_BASE(cp_fmt) <= MMSA_FMT_F3, "gemm_split3: bad plane format %d / %d", fmt, cp_fmt);
  MMSA_CHECK_ARG(MMSA_CP_SPLIT(cp_fmt) == 0 || (out_mode == 0 && MMSA_CP_SPLIT(cp_fmt) < N && MMSA_CP_BASE(cp_fmt) != MMSA_FMT_H8C), "gemm_split3: the output-format split %d needs a plain [M, N] bf16 hi/lo planes output with N=%d beyond it", MMSA_CP_SPLIT(cp_fmt), N);
  MMSA_CHECK_ARG(fmt == MMSA_FMT_B3 || fmt == MMSA_FMT_F3 || (ap && K % 64 == 0), "gemm_split3: h8 / h8c operands need A planes and K %% 64 == 0 (K=%d)", K);
  MMSA_CHECK_ARG(fmt != MMSA_FMT_F3 || ap || !Cp, "gemm_split3: f3 weights with an fp32 A write fp32 outputs only");
  const bool h8c = fmt == MMSA_FMT_H8C, cp_h8c = MMSA_CP_BASE(cp_fmt) == MMSA_FMT_H8C;
  MMSA_CHECK_ARG((A || Ap) && Wp && (C || Cp), "gemm_split3: null pointer");
  MMSA_CHECK_ARG(!(A && Ap), "gemm_split3: pass either fp32 A or A planes, not both");
  MMSA_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0, "gemm_split3: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  MMSA_CHECK_ARG(K % BK == 0, "gemm_split3: K=%d must be a multiple of %d (producers pad)", K, BK);
  if (ap) {
    MMSA_CHECK_ARG((lda & 63) == 0 && (strideA & 63) == 0 && (((uintptr_t)Ap) & 127) == 0,
                   "gemm_split3: A planes must be 128-byte aligned with lda%%64==0 (lda=%ld)", lda);
    MMSA_CHECK_ARG(lda >= (h8c ? 3L : 2L) * K, "gemm_split3: planes lda=%ld < %d*K=%d (h8c: lda is the row-PAIR stride)", lda, h8c ? 3 : 2, K);
  } else {
    MMSA_CHECK_ARG((lda & 3) == 0 && (strideA & 3) == 0 && (((uintptr_t)A) & 15) == 0, "gemm_split3: A must be 16-byte aligned with lda%%4==0 (lda=%ld)", lda);
  }
  MMSA_CHECK_ARG(lda >= K, "gemm_split3: lda=%ld < K=%d", lda, K);
  MMSA_CHECK_ARG((((uintptr_t)Wp) & 127) == 0 && (strideW & 63) == 0, "gemm_split3: weight planes must be 128-byte aligned");
  MMSA_CHECK_ARG((!C || (((uintptr_t)C) & 15) == 0) && (!resid || (((uintptr_t)resid) & 15) == 0), "gemm_split3: C/resid must be 16-byte aligned");
  MMSA_CHECK_ARG(!Cp || ((((uintptr_t)Cp) & 127) == 0 && (ldcp & 63) == 0 && (strideCp & 63) == 0), "gemm_split3: output planes must be 128-byte aligned, ldcp%%64==0");
  MMSA_CHECK_ARG(act >= ACT_NONE && act <= ACT_SIGMOID, "gemm_split3: bad act %d", act);
  if (out_mode == 1) {
    MMSA_CHECK_ARG(ps_H > 0 && ps_W > 0 && ps_C > 0 && N == 4 * ps_C && M % (ps_H * ps_W) == 0 && (ps_C & 3) == 0,
                   "gemm_split3: pixel-shuffle store needs N==4*C, M%%(H*W)==0");
    MMSA_CHECK_ARG((!C || ldc >= ps_C) && (!Cp || ldcp >= (cp_h8c ? 3L * MMSA_PAD64(ps_C) : 2L * ps_C)), "gemm_split3: ldc < C");
  } else {
    MMSA_CHECK_ARG(out_mode == 0, "gemm_split3: bad out_mode %d", out_mode);
    MMSA_CHECK_ARG((!C || ldc >= N) && (!Cp || ldcp >= (cp_h8c ? 3L * MMSA_PAD64(N) : 2L * ((N + 31) / 32 * 32))), "gemm_split3: ldc=%ld < N=%d", ldc, N);
  }
  GemmArgs a;
  a.A = A; a.Ap = Ap; a.lda = lda; a.strideA = strideA;
  a.Wp = Wp; a.strideW = strideW;
  a.bias = bias; a.strideBias = strideBias; a.colscale = colscale;
  a.resid = resid; a.ldr = ldr; a.strideR = strideR; a.resid_mod = resid_mod; a.beta = beta;
  a.C = C; a.ldc = C ? ldc : 0; a.strideC = strideC;
  a.Cp = Cp; a.ldcp = Cp ? ldcp : 0; a.strideCp = strideCp;
  a.M = M; a.N = N; a.K = K; a.act = act; a.alpha = alpha;
  a.out_mode = out_mode; a.ps_H = ps_H; a.ps_W = ps_W; a.ps_C = ps_C; a.cp_fmt = cp_fmt;
  a.clamp_max = clamp_max;
  // main path: activation planes go to the LDS-DMA / 256x128 kernel (gemm_v2.hip); MMSA_GEMM_V1=1 (debug-knob builds) forces this one
  const bool force_v1 = MMSA_KNOB("MMSA_GEMM_V1", 0) != 0;
  // (round 1 routed one-strip shapes with many rows and a deep K -- ConvNeXt stage-0 pw2: N = 96, K = 384 -- to the 128-row tiles of
  // this kernel; with the blocked tile order and the 96-column tiles of the LDS-DMA kernel that shape is 18 % faster there: 152 -> 125 us)
  const bool narrow = false;
  const bool no_tiny = MMSA_KNOB("MMSA_GEMM_NO_TINY", 0) != 0;   // A/B aid (debug-knob builds)
  // routed by the problem's small dimension, NOT by the row count (rows scale with the image batch: a batch-dependent choice of
  // kernel would make results depend on how images are batched); M <= 16384 covers 32 images of the largest pooled map
  if (!ap && !Cp && out_mode == 0 && resid_mod <= 0 && (N <= 64 || K <= 64) && M <= 16384 && !no_tiny && !extras && fmt == MMSA_FMT_B3) {
    const int ncg = cdiv(N, 8);
    hipLaunchKernelGGL(gemm_tiny_kernel, dim3(cdiv((long)M * ncg, 4), batch), dim3(256), 0, stream, a, ncg);
    MMSA_CHECK_LAUNCH("gemm_split3(tiny)");
    return MMSA_OK;
  }
  if (ap && (fmt != MMSA_FMT_B3 || (M >= 128 && !force_v1 && !narrow))) {   // h8 / h8c operands: only the LDS-DMA kernels read them
    return mmsa_gemm_v2_launch(Ap, lda, strideA, Wp, strideW, bias, strideBias, colscale, resid, ldr, strideR,
                               resid_mod, beta, C, ldc, strideC, Cp, ldcp, strideCp, M, N, K, batch, act, alpha,
                               out_mode, ps_H, ps_W, ps_C, fmt, cp_fmt, max_grid, stream, rs_out, rn_mr, rn_cs, flavour, clamp_max);
  }
  MMSA_CHECK_ARG(!extras, "gemm_split3: this shape is not routed to the LDS-DMA kernel, which alone writes row statistics / normalises rows");
  MMSA_CHECK_ARG(!cp_h8c, "gemm_split3: h8c output planes are written by the LDS-DMA kernel only (activation planes in, M >= 128)");
  dim3 grid(cdiv(N, BN), cdiv(M, BM), batch);
  const bool gen = out_mode != 0 || resid_mod > 0;
  if (ap) {
    if (gen) hipLaunchKernelGGL((gemm_split3_kernel<true, true>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_split3_kernel<true, false>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
  } else if (fmt == MMSA_FMT_F3) {
    if (gen) hipLaunchKernelGGL((gemm_split3_kernel<false, true, true>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_split3_kernel<false, false, true>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
  } else {
    if (gen) hipLaunchKernelGGL((gemm_split3_kernel<false, true>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
    else hipLaunchKernelGGL((gemm_split3_kernel<false, false>), grid, dim3(256), 2 * STAGE_BYTES, stream, a);
  }
  MMSA_CHECK_LAUNCH("gemm_split3");
  return MMSA_OK;
}

// ---- pre-pack: fp32 [rows, cols] (row stride ld) -> planes [rows, 2*cols_pad] (zero padded).
// kind 0: bf16 hi/lo planes; 1: h8 activation rows (chunk = lo bytes | q(hi) bytes); 2: h8 WEIGHT rows (chunk = q(hi) | lo); 3: h8c planes; 4: f3 (fp16 hi/lo)
// (dense: row-pair stride 3 * cols_pad; rows odd: the pair partner of the last row is not written): common.h
__global__ void split_planes_kernel(const float* __restrict__ src, long ld, int rows, int cols, int cols_pad,
                                    unsigned short* __restrict__ out, int kind, float* __restrict__ clamp_max) {
  const long total = (long)rows * cols_pad;
  float cw_ = 0.f;   // clamp watch (common.h)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
    const float x = c < cols ? src[(long)r * ld + c] : 0.f;
    clamp_see1(cw_, x);
    if (kind == 3) {   // h8c: row pairs [hi row 2j | hi row 2j+1 | lo lines] (common.h)
      h8c_store1(h8c_row(out, 3L * cols_pad, r, cols_pad), c, x);
      continue;
    }
    unsigned short* row = out + (long)r * 2 * cols_pad;
    if (kind == 0) {
      unsigned short h, l;
      split_bf16(x, h, l);
      row[ilv(c)] = h;
      row[ilv(c) + 32] = l;
    } else if (kind == 4) {   // f3: the same layout with fp16 halves
      store_planes1(row, c, x, MMSA_FMT_F3);
    } else {
      unsigned hi, lo8 = 0u, qh8 = 0u;
      h8_split2<false>(x, 0.f, hi, lo8, qh8);
      row[ilv(c)] = (unsigned short)(hi & 0xFFFFu);
      unsigned char* rb = reinterpret_cast<unsigned char*>(row) + h8_lo_off(c & ~3) + (c & 3);
      rb[kind == 2 ? 8 : 0] = (unsigned char)(lo8 & 0xFFu);
      rb[kind == 2 ? 0 : 8] = (unsigned char)(qh8 & 0xFFu);
    }
  }
  clamp_report(clamp_max, cw_, kind == 0 ? 3.0e38f : kind == 4 ? MMSA_F3_MAX : MMSA_H8_MAX);
}

extern "C" int mmsa_split_planes(const float* src, long ld, int rows, int cols, int cols_pad,
                                 unsigned short* out, int kind, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(src && out && rows > 0 && cols > 0 && cols_pad >= cols && cols_pad % 32 == 0, "split_planes: bad args");
  MMSA_CHECK_ARG(kind >= 0 && kind <= 4, "split_planes: kind %d (0 bf16 hi/lo, 1 h8 activation, 2 h8 weight, 3 h8c, 4 f3)", kind);
  MMSA_CHECK_ARG(kind != 3 || cols_pad % 64 == 0, "split_planes: h8c planes need cols_pad %% 64 == 0");
  const long total = (long)rows * cols_pad;
  int blocks = cdiv(total, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, stream, src, ld, rows, cols, cols_pad, out, kind, clamp_max);
  MMSA_CHECK_LAUNCH("split_planes");
  return MMSA_OK;
}
