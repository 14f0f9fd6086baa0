// Streaming GEMM for the skinny launches of the neck (round 6; VERDICT r05 "next" item 6b): C[M, N] = act(A[M, K] W[N, K]^T + bias) * scale + beta * resid with
// M in the tens of thousands and N, K <= 192 -- MobileNetV2's 1 x 1 convs at the finest level (AM:281-295: c -> 2c with ReLU6, 2c -> c with the residual), 131072
// rows at two images.  These launches move 150-250 MB for 5 GFLOP: 20-30 FLOP per byte against a machine balance of 312.  On the tiled LDS-DMA kernel
// (gemm_v2.hip, 128-row tiles, two workgroups per CU) they ran at 1.6-2.4 TB/s: every tile re-streams the whole weight matrix through the ring beside its A
// rows (a third to three fifths of the DMA lines), a tile is 3-6 k-steps of prologue, and its epilogue is four serialised sub-tiles.
//
// Here nothing is tiled over N or K.  One workgroup of 8 waves per CU keeps the WHOLE weight matrix in LDS for the life of the kernel (<= 147 KiB; row stride
// padded by 16 bytes so that the 16 rows of a fragment read fall on different 16-byte slots); every wave then works ALONE -- no barrier after the fill -- on
// 16-row blocks of A: its lanes load their MFMA operand fragments of the block straight from global memory into registers (lane (l15, g): row l15, the 8 hi and
// 8 lo values of k-chunk g of every 32-wide k-block: 2 KS loads of 16 bytes), the NEXT block's loads are issued before the current block's MFMAs, and the
// results leave from the accumulator layout, which for W . A^T holds four consecutive columns of one row per lane (16-byte stores / plane pieces).  Per CU
// 8 waves x 2 blocks x up to 12 KiB are in flight: enough to cover HBM latency without any shared staging.
// Arithmetic: the same three MFMAs per k-block in the same order as gemm_v2 (lo.hi, hi.lo, hi.hi; k ascending) and the same epilogue operations in the same
// order, so a launch routed here returns what the tiled kernel returns (tests/test_planes_gpu.py::test_gemm_stream_kernel).
#include "gemm_v2_shared.h"

struct GemmStreamArgs {
  const unsigned short* Ap; long lda;     // A planes (bf16 hi/lo or f3), row stride in uint16
  const unsigned short* Wp;               // W planes [N, 2K] dense
  const float* bias; const float* colscale; float alpha;
  const float* resid; long ldr; float beta;
  float* C; long ldc;
  unsigned short* Cp; long ldcp; int cp_fmt;
  int M, act;
  float* clamp_max;
};

#define GS_WROW(K_) ((K_) * 4 + 16)       // bytes per W row in LDS (128 bytes per k-block + 16 of padding per row)

template <int NT, int KS, bool F16, bool RES>
__global__ __launch_bounds__(512, 1) void gemm_stream_kernel(GemmStreamArgs a) {
  constexpr int N = NT * 16, K = KS * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;

  // ---- the weight matrix into LDS, once: N rows x K / 32 lines of 128 bytes, 16-byte pieces
  {
    constexpr int PIECES = N * KS * 8;
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.Wp);
    for (int p = tid; p < PIECES; p += 512) {
      const int n = p / (KS * 8), q = p - n * (KS * 8);
      *reinterpret_cast<uint4*>(smem + n * GS_WROW(K) + q * 16) = *reinterpret_cast<const uint4*>(wsrc + (long)n * (K * 4) + q * 16);
    }
  }
  if (tid < N) {
    float* cvl = reinterpret_cast<float*>(smem + N * GS_WROW(K));
    cvl[tid] = a.bias ? a.bias[tid] : 0.f;
    cvl[N + tid] = (a.colscale ? a.colscale[tid] : 1.f) * a.alpha;
  }
  __syncthreads();

  const int nblk = (a.M + 15) >> 4;
  const int stride = gridDim.x * 8;
  int blk = blockIdx.x * 8 + wave;
  if (blk >= nblk) return;                  // (after the only barrier of the kernel)
  const unsigned char* Ab = reinterpret_cast<const unsigned char*>(a.Ap);
  const long ldaB = a.lda * 2;
  const unsigned woff0 = (unsigned)(l15 * GS_WROW(K) + g * 16);   // this lane's piece of W row l15 of an n-tile: + nt * 16 rows, + ks * 128, lo: + 64

  uint4 fa[2][KS][2];                       // [buffer][k-block][hi | lo]: this lane's A fragments of a 16-row block
  auto load_block = [&](int b_, uint4 (&f)[KS][2]) {
    const int row = min(b_ * 16 + l15, a.M - 1);            // rows beyond M: clamped (never stored)
    const unsigned char* rp = Ab + (long)row * ldaB + g * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f[ks][0] = *reinterpret_cast<const uint4*>(rp + ks * 128);
      f[ks][1] = *reinterpret_cast<const uint4*>(rp + ks * 128 + 64);
    }
  };
  const int cp_base = MMSA_CP_BASE(a.cp_fmt);
  const bool act_on = a.act != ACT_NONE;
  const float act_hi = a.act == ACT_RELU6 ? 6.0f : INFINITY;
  float cw_ = 0.f;                          // clamp watch (common.h): the largest |value| this lane converted to f3 planes
  // per-column vectors (bias, scale = colscale * alpha) live in LDS behind the weight image (filled before the barrier above): held in registers they
  // cost 8 NT VGPRs and pushed the kernel into scratch
  const float* colv = reinterpret_cast<const float*>(smem + N * GS_WROW(K)) + 4 * g;   // bias: + 16 nt; scale: + N + 16 nt

  auto compute_store = [&](int b_, const uint4 (&f)[KS][2]) {
    // the block's residual rows first: they land under the MFMAs (lane = row l15, columns 16 nt + 4 g .. + 3, as the results will be held)
    const int m = b_ * 16 + l15;
    float4 rr[NT];
    if constexpr (RES) {      // (a compile-time flag: a run-time branch around the loads makes the compiler drain the whole queue at the join)
      const float* rrow = a.resid + (long)min(m, a.M - 1) * a.ldr + 4 * g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rr[nt] = *reinterpret_cast<const float4*>(rrow + nt * 16);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the weight fragments are the same for every block: seen through an opaque offset, or the compiler hoists all 2 NT KS of them out of the block loop
    // (288 registers at N = 96, K = 192: the first build spilled 870 bytes per lane)
    unsigned woff = woff0;
    asm volatile("" : "+v"(woff));
    const unsigned char* wl_ = smem + woff;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const uint4 wh = *reinterpret_cast<const uint4*>(wl_ + nt * 16 * GS_WROW(K) + ks * 128);
        const uint4 wl = *reinterpret_cast<const uint4*>(wl_ + nt * 16 * GS_WROW(K) + ks * 128 + 64);
        if constexpr (F16) {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, f[ks][0]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, f[ks][1]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, f[ks][0]), acc[nt], 0, 0, 0);
        } else {
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, f[ks][0]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, f[ks][1]), acc[nt], 0, 0, 0);
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, f[ks][0]), acc[nt], 0, 0, 0);
        }
      }
    }
    // ---- epilogue from the accumulator layout: lane = row l15 of the block, columns 16 nt + 4 g .. + 3
    if (m < a.M) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + 4 * g;
        const float4 bv = *reinterpret_cast<const float4*>(colv + 16 * nt), cv = *reinterpret_cast<const float4*>(colv + N + 16 * nt);
        float4 o = make_float4(acc[nt][0] + bv.x, acc[nt][1] + bv.y, acc[nt][2] + bv.z, acc[nt][3] + bv.w);
        // ReLU / ReLU6 as fmin(fmax(x, 0), hi) with hi = 6 or +inf: the operations of apply_act (common.h) without a switch per element
        if (act_on) { o.x = fminf(fmaxf(o.x, 0.0f), act_hi); o.y = fminf(fmaxf(o.y, 0.0f), act_hi); o.z = fminf(fmaxf(o.z, 0.0f), act_hi); o.w = fminf(fmaxf(o.w, 0.0f), act_hi); }
        o.x *= cv.x; o.y *= cv.y; o.z *= cv.z; o.w *= cv.w;
        if constexpr (RES) { o.x += a.beta * rr[nt].x; o.y += a.beta * rr[nt].y; o.z += a.beta * rr[nt].z; o.w += a.beta * rr[nt].w; }
        if (a.C) *reinterpret_cast<float4*>(a.C + (long)m * a.ldc + n) = o;
        if (a.Cp) {
          clamp_see(cw_, o);
          store_planes4(a.Cp + (long)m * a.ldcp, n, o, cp_base);
        }
      }
    }
  };

  // two blocks in flight per wave; every load is UNCONDITIONAL (a block index past the end is clamped to the last block: a redundant load, never a branch --
  // hipcc waits vmcnt(0) behind any load it had to branch around, which would drain the prefetch)
  const int lastb = nblk - 1;
  load_block(blk, fa[0]);
#pragma unroll 1
  for (;;) {
    const int nb1 = blk + stride;
    load_block(min(nb1, lastb), fa[1]);
    compute_store(blk, fa[0]);
    if (nb1 >= nblk) break;
    const int nb2 = nb1 + stride;
    load_block(min(nb2, lastb), fa[0]);
    compute_store(nb1, fa[1]);
    if (nb2 >= nblk) break;
    blk = nb2;
  }
  if (a.Cp) clamp_report(a.clamp_max, cw_, mmsa_clamp_limit(cp_base));
}

// Which launches take this kernel (called by mmsa_gemm_v2_launch before it builds its own argument block): both operands as bf16 hi/lo or f3 planes, one batch,
// (N, K) = (96, 192) or (192, 96), plain row mapping, no LayerNorm-fold outputs, a planes output (if any) in the operands' pair family, everything 16-byte aligned, and enough
// rows that the launch is a stream (a tile-sized problem keeps the tiled kernel).  Returns 1 when it launched, 0 when the shape is not its, < 0 on a launch error.
int mmsa_gemm_stream_try(const unsigned short* Ap, long lda, const unsigned short* Wp, const float* bias, const float* colscale, const float* resid, long ldr, float beta,
                         float* C, long ldc, unsigned short* Cp, long ldcp, int M, int N, int K, int batch, int act, float alpha, int out_mode, int resid_mod,
                         int fmt, int cp_fmt, int max_grid, const float* rs_out, const float* rn_mr, int flavour, float* clamp_max, hipStream_t stream) {
#ifndef MMSA_GEMM_STREAM_DEFAULT
#define MMSA_GEMM_STREAM_DEFAULT 1   // 0 (A/B builds: tools/build_variant.sh ... gemm_stream.hip -DMMSA_GEMM_STREAM_DEFAULT=0): every launch keeps the tiled kernel
#endif
  if (MMSA_KNOB("MMSA_GEMM_STREAM", MMSA_GEMM_STREAM_DEFAULT) == 0 || flavour != 0) return 0;
  if (!(fmt == MMSA_FMT_B3 || fmt == MMSA_FMT_F3) || batch != 1 || out_mode != 0 || resid_mod > 0 || rs_out || rn_mr) return 0;
  // the two shapes it wins on (profiles/r06_gemm_stream.txt: 2.46 x and 1.39 x against the tiled kernel; 192 x 192 ties and 96 x 96 loses 7 %, so those keep the tiled kernel)
  if (!((N == 96 && K == 192) || (N == 192 && K == 96)) || M < 16384) return 0;
  if (!(act == ACT_NONE || act == ACT_RELU || act == ACT_RELU6)) return 0;
  if (Cp && (MMSA_CP_SPLIT(cp_fmt) != 0 || !(MMSA_CP_BASE(cp_fmt) == MMSA_FMT_B3 || MMSA_CP_BASE(cp_fmt) == MMSA_FMT_F3))) return 0;
  const uintptr_t al = (uintptr_t)Ap | (uintptr_t)Wp | (uintptr_t)bias | (uintptr_t)colscale | (uintptr_t)resid | (uintptr_t)C | (uintptr_t)Cp;
  if ((al & 15) || (lda & 7) || (C && (ldc & 3)) || (resid && (ldr & 3)) || (Cp && (ldcp & 7)) || lda < 2L * K || (Cp && ldcp < 2L * N)) return 0;
  GemmStreamArgs a;
  a.Ap = Ap; a.lda = lda; a.Wp = Wp; a.bias = bias; a.colscale = colscale; a.alpha = alpha; a.resid = resid; a.ldr = ldr; a.beta = beta;
  a.C = C; a.ldc = ldc; a.Cp = Cp; a.ldcp = ldcp; a.cp_fmt = cp_fmt; a.M = M; a.act = act; a.clamp_max = clamp_max;
  static MmsaPerDevice per_dev_ = {};
  const int num_cus = mmsa_per_device(per_dev_, [] {
#define GS_ATTR(NT_, KS_, F_) (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<NT_, KS_, F_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (NT_) * 16 * GS_WROW((KS_) * 32) + (NT_) * 16 * 8); \
                              (void)hipFuncSetAttribute((const void*)gemm_stream_kernel<NT_, KS_, F_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (NT_) * 16 * GS_WROW((KS_) * 32) + (NT_) * 16 * 8);
    GS_ATTR(6, 6, false) GS_ATTR(6, 6, true) GS_ATTR(12, 3, false) GS_ATTR(12, 3, true)
#undef GS_ATTR
  });
  const int cus = (max_grid > 0 && max_grid < num_cus) ? max_grid : num_cus;
  const int nblk = (M + 15) / 16;
  const int grid = nblk / 8 < cus ? (nblk + 7) / 8 : cus;
  const int lds = N * GS_WROW(K) + N * 8;
  const bool f16 = fmt == MMSA_FMT_F3;
#define GS_LAUNCH(NT_, KS_)                                                                                                   \
  { if (f16 && resid) hipLaunchKernelGGL((gemm_stream_kernel<NT_, KS_, true, true>), dim3(grid), dim3(512), lds, stream, a);      \
    else if (f16) hipLaunchKernelGGL((gemm_stream_kernel<NT_, KS_, true, false>), dim3(grid), dim3(512), lds, stream, a);          \
    else if (resid) hipLaunchKernelGGL((gemm_stream_kernel<NT_, KS_, false, true>), dim3(grid), dim3(512), lds, stream, a);        \
    else hipLaunchKernelGGL((gemm_stream_kernel<NT_, KS_, false, false>), dim3(grid), dim3(512), lds, stream, a); }
  if (N == 96 && K == 192) GS_LAUNCH(6, 6)
  else GS_LAUNCH(12, 3)
#undef GS_LAUNCH
  MMSA_CHECK_LAUNCH("gemm_split3(stream)");
  return 1;
}
