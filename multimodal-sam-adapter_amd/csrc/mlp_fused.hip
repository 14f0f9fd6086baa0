// Fused ConvNeXt pointwise pair for the narrowest stage (C = 96):
//   x <- x + gamma * (GELU(A W1^T + b1) W2^T + b2)            (TC:107-132: pointwise_conv1 -> GELU -> pointwise_conv2 -> *gamma -> +res)
// A = LayerNorm output as bf16 hi/lo planes [rows, C]; W1 [4C, C], W2 [C, 4C] as bf16 hi/lo planes; x fp32 [rows, C].
//
// Why.  At C = 96 the two GEMMs have 3 k-tiles per output tile against an epilogue that writes (pw1) and an operand stream that
// re-reads (pw2) the 4C-wide hidden tensor -- 403 MB per block at stage 0, batch 2: both launches are bound by their epilogues /
// operand staging, not by the matrix pipe (LAB_NOTES.md 4.1).  Here a workgroup keeps the A image of a 128-row tile in LDS (48 KiB)
// and walks the hidden dimension in chunks of 64 columns, two steps per chunk: step A computes the chunk's hidden values (K = C: the
// whole W1 slice of the chunk is ONE 24 KiB ring slot), applies bias + GELU, splits them and writes them as an A-operand image into
// LDS; step B consumes that image as the K = 64 slice of the second contraction (the W2 slice of the chunk is again one 24 KiB slot)
// into accumulators that stay in registers for the whole tile.  The hidden tensor never leaves the CU.
//
// 8 waves = 4 (32-row strips) x 2 (column halves); a ring of 3 slots fetched two steps ahead by LDS-DMA (3 pieces per wave and
// step, counted vmcnt), one barrier per step; the next tile's A image is requested under the last step and the epilogue of the
// current one.  Same swizzled 128-byte-row LDS images, fragment reads and bf16 hi/lo product (3 MFMAs) as gemm_v2.hip: the results
// are those of the two separate GEMMs up to the summation order of the second contraction (64-column chunks in order).
// C = 192 (stage 1) was measured with a k-tile-per-step version of this kernel: slower than the pair of launches (217 vs 185 us),
// whose 256-row tiles reuse every weight byte twice as often; it stays on the two GEMMs.
#include "common.h"

#define MF_GLDS16(gptr, lptr)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct MlpFusedArgs {
  const unsigned short* Ap; long lda; long strideA;      // activation planes, row stride lda (bf16 units, >= 2C), batch stride
  const unsigned short* W1p; long strideW1;              // [4C, 2C] planes per batch
  const unsigned short* W2p; long strideW2;              // [C, 8C] planes per batch
  const float* b1; const float* b2; const float* gamma;  // [batch, 4C], [batch, C], [batch, C]
  float* x; long ldx; long strideX;                      // residual in / result out, fp32
  int M, batch;                                          // rows per batch
  int ntiles, tiles_per_batch;
  float* clamp;                                          // optional clamp watch word (common.h): the hidden tensor GELU(A W1^T + b1) is converted to planes in the kernel
};

#define MF_C 96
#define MF_HID (4 * MF_C)
#define MF_AS_BYTES (3 * 16384)                         // A image: 3 k-blocks x 128 rows x 128 B
#define MF_SLOT 24576                                   // W1 slice (3 k-blocks x 64 rows) or W2 slice (2 k-blocks x 96 rows)
#define MF_RING_OFF MF_AS_BYTES
#define MF_HS_OFF (MF_RING_OFF + 3 * MF_SLOT)           // hidden image: 2 k-blocks x 128 rows x 128 B
#define MF_BS_OFF (MF_HS_OFF + 32768)                   // b1 of every batch (<= 2048 floats)
#define MF_LDS (MF_BS_OFF + 8192)                       // 160 KiB

// F16: A, W1, W2 and the hidden image are fp16 hi/lo pairs ("f3" planes, common.h) and the products run on the fp16 MFMA; else bf16 hi/lo
template <bool F16>
__global__ __launch_bounds__(512, 1) void mlp_fused_kernel(MlpFusedArgs a) {
  float cw_ = 0.f;   // clamp watch: the largest |hidden value| this lane converted to f3 planes (reported once, at the end)
  constexpr int C = MF_C, HID = MF_HID;
  constexpr int NSTEP = 2 * (HID / 64);   // 12 steps per tile: (A, B) per 64-column hidden chunk
  constexpr int NT = C / 32;              // 16-column output tiles per wave (C / 2 columns per wave)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Hs = smem + MF_HS_OFF;
  float* Bs = reinterpret_cast<float*>(smem + MF_BS_OFF);   // b1 of every batch: read per chunk from LDS (a global load inside the step loop
                                                            // would sit in the in-order vmcnt queue behind the DMA pieces in flight)
  for (int i = threadIdx.x; i < a.batch * HID; i += 512) Bs[i] = a.b1[i];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int drow = lane >> 3;
  const int dp_even = ((lane & 7) ^ (drow >> 1)) * 8;     // source piece (bf16 units) of an even 8-row group; odd groups: ^ 32 (gemm_v2.hip)
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag_hi = l15 * 128 + fslot * 16;
  const int frag_lo = l15 * 128 + (fslot ^ 4) * 16;
  // this wave's W1 rows (8 wave .. 8 wave + 7 of a chunk's 64) and W2 pieces are fixed for the kernel: row offsets in bf16 units
  const long w1_off = (long)(8 * wave + drow) * (2 * C) + ((wave & 1) ? (dp_even ^ 32) : dp_even);
  int w2_kb[3], w2_dst[3];
  long w2_off[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int p = wave + 8 * i, kb2 = p / 12, q = p - 12 * kb2;          // piece p of 24: k-block kb2 of the chunk, rows 8q .. 8q+7 of W2
    w2_kb[i] = kb2;
    w2_dst[i] = kb2 * 12288 + q * 1024;
    w2_off[i] = (long)(8 * q + drow) * (2 * HID) + ((q & 1) ? (dp_even ^ 32) : dp_even);
  }

  auto tile_of = [&](int tile, int& bz, int& m0) { bz = tile / a.tiles_per_batch; m0 = (tile - bz * a.tiles_per_batch) * 128; };
  auto issue_a = [&](int tile) {           // the tile's A image: 48 pieces, 6 per wave
    int bz, m0;
    tile_of(tile, bz, m0);
    const unsigned short* Ab = a.Ap + (long)bz * a.strideA;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int p = wave + 8 * i, kb = p >> 4, q = p & 15;
      const int row = min(m0 + 8 * q + drow, a.M - 1);
      MF_GLDS16(Ab + (long)row * a.lda + kb * 64 + ((q & 1) ? (dp_even ^ 32) : dp_even), As + kb * 16384 + q * 1024);
    }
  };
  auto issue = [&](int t, const unsigned short* W1b, const unsigned short* W2b, int slot) {   // step t of a tile: 3 pieces per wave
    unsigned char* sb = smem + MF_RING_OFF + slot * MF_SLOT;
    const int j = t >> 1;
    if ((t & 1) == 0) {
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) MF_GLDS16(W1b + (long)j * 64 * (2 * C) + w1_off + kb * 64, sb + kb * 8192 + wave * 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i) MF_GLDS16(W2b + w2_off[i] + (2 * j + w2_kb[i]) * 64, sb + w2_dst[i]);
    }
  };

  int slot_c = 0;                                         // ring slot of the step being computed (runs on across tiles)
  int tile = blockIdx.x;
  if (tile < a.ntiles) issue_a(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    int bz, m0;
    tile_of(tile, bz, m0);
    const unsigned short* W1b = a.W1p + (long)bz * a.strideW1;
    const unsigned short* W2b = a.W2p + (long)bz * a.strideW2;
    f32x4 acc_o[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < NT; ++nj) acc_o[mi][nj] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc_h[2][2];
    issue(0, W1b, W2b, slot_c);
    issue(1, W1b, W2b, slot_c == 2 ? 0 : slot_c + 1);

#pragma unroll 1
    for (int t = 0; t < NSTEP; ++t) {
      // step t (and everything older: the A image, the previous tile's stores) landed; the 3 pieces of t + 1 may stay in flight
      if (t + 1 < NSTEP) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __syncthreads();
      const unsigned char* sb = smem + MF_RING_OFF + slot_c * MF_SLOT;
      const int pf_slot = slot_c == 0 ? 2 : slot_c - 1;    // slot of step t + 2 = slot of step t - 1: free after the barrier above
      const int j = t >> 1;
      if ((t & 1) == 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc_h[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
          bf16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(As + kb * 16384 + (wm * 32 + i * 16) * 128 + frag_hi);
            al[i] = *reinterpret_cast<const bf16x8*>(As + kb * 16384 + (wm * 32 + i * 16) * 128 + frag_lo);
            wh[i] = *reinterpret_cast<const bf16x8*>(sb + kb * 8192 + (wn * 32 + i * 16) * 128 + frag_hi);
            wl[i] = *reinterpret_cast<const bf16x8*>(sb + kb * 8192 + (wn * 32 + i * 16) * 128 + frag_lo);
          }
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              acc_h[mi][ni] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl[ni]), __builtin_bit_cast(f16x8, ah[mi]), acc_h[mi][ni], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni], ah[mi], acc_h[mi][ni], 0, 0, 0));
              acc_h[mi][ni] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ni]), __builtin_bit_cast(f16x8, al[mi]), acc_h[mi][ni], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], al[mi], acc_h[mi][ni], 0, 0, 0));
              acc_h[mi][ni] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh[ni]), __builtin_bit_cast(f16x8, ah[mi]), acc_h[mi][ni], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], ah[mi], acc_h[mi][ni], 0, 0, 0));
            }
        }
        if (t + 2 < NSTEP) issue(t + 2, W1b, W2b, pf_slot);
        // h = GELU(acc + b1) -> bf16 hi/lo -> A-operand image of k-block `wn` (this wave's 32 hidden columns of the chunk) in Hs.
        // Lane: row = wm*32 + mi*16 + l15, hidden columns (within the k-block) ni*16 + 4g .. +3.
        const float* b1p = Bs + bz * HID + j * 64 + wn * 32;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float4 bb = *reinterpret_cast<const float4*>(b1p + ni * 16 + 4 * g);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            float4 o = make_float4(acc_h[mi][ni][0] + bb.x, acc_h[mi][ni][1] + bb.y, acc_h[mi][ni][2] + bb.z, acc_h[mi][ni][3] + bb.w);
            o = gelu4(o);
            uint2 hh, ll;
            if constexpr (F16) { clamp_see(cw_, o); f3_split4(o, hh, ll); } else split4(o, hh, ll);
            const int row = wm * 32 + mi * 16 + l15;
            const int chunk = ni * 2 + (g >> 1);                      // 16-byte chunk (8 values) of the 32-wide k-block
            const int sw = (row >> 1) & 7;
            unsigned char* rb = Hs + wn * 16384 + row * 128 + (g & 1) * 8;
            *reinterpret_cast<uint2*>(rb + ((chunk ^ sw) << 4)) = hh;
            *reinterpret_cast<uint2*>(rb + (((chunk + 4) ^ sw) << 4)) = ll;
          }
        }
      } else {
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          bf16x8 ah[2], al[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            ah[i] = *reinterpret_cast<const bf16x8*>(Hs + kb2 * 16384 + (wm * 32 + i * 16) * 128 + frag_hi);
            al[i] = *reinterpret_cast<const bf16x8*>(Hs + kb2 * 16384 + (wm * 32 + i * 16) * 128 + frag_lo);
          }
#pragma unroll
          for (int nj = 0; nj < NT; ++nj) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(sb + kb2 * 12288 + (wn * (C / 2) + nj * 16) * 128 + frag_hi);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(sb + kb2 * 12288 + (wn * (C / 2) + nj * 16) * 128 + frag_lo);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
              acc_o[mi][nj] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, ah[mi]), acc_o[mi][nj], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, ah[mi], acc_o[mi][nj], 0, 0, 0));
              acc_o[mi][nj] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, al[mi]), acc_o[mi][nj], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, al[mi], acc_o[mi][nj], 0, 0, 0));
              acc_o[mi][nj] = (F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, ah[mi]), acc_o[mi][nj], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mi], acc_o[mi][nj], 0, 0, 0));
            }
          }
        }
        if (t + 2 < NSTEP) issue(t + 2, W1b, W2b, pf_slot);
        else if (t == NSTEP - 1 && tile + (int)gridDim.x < a.ntiles) issue_a(tile + gridDim.x);   // every wave is past its last read of As (step 10): the next tile's image
      }
      slot_c = slot_c == 2 ? 0 : slot_c + 1;
    }

    // ---- tile epilogue: x = x + gamma * (acc + b2); lane: row = wm*32 + mi*16 + l15, columns wn*(C/2) + nj*16 + 4g .. +3
    float* xb = a.x + (long)bz * a.strideX;
    const float* b2p = a.b2 + (long)bz * C + wn * (C / 2);
    const float* gp = a.gamma + (long)bz * C + wn * (C / 2);
#pragma unroll
    for (int nj = 0; nj < NT; ++nj) {
      const float4 bb = *reinterpret_cast<const float4*>(b2p + nj * 16 + 4 * g);
      const float4 gg = *reinterpret_cast<const float4*>(gp + nj * 16 + 4 * g);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + wm * 32 + mi * 16 + l15;
        if (m < a.M) {
          float* px = xb + (long)m * a.ldx + wn * (C / 2) + nj * 16 + 4 * g;
          const float4 rr = *reinterpret_cast<const float4*>(px);
          float4 o;
          o.x = rr.x + gg.x * (acc_o[mi][nj][0] + bb.x);
          o.y = rr.y + gg.y * (acc_o[mi][nj][1] + bb.y);
          o.z = rr.z + gg.z * (acc_o[mi][nj][2] + bb.z);
          o.w = rr.w + gg.w * (acc_o[mi][nj][3] + bb.w);
          *reinterpret_cast<float4*>(px) = o;
        }
      }
    }
  }
  if constexpr (F16) clamp_report(a.clamp, cw_, MMSA_F3_MAX);
}

// x[b] <- x[b] + gamma[b] * (GELU(A[b] W1[b]^T + b1[b]) W2[b]^T + b2[b]) for b < batch; C = 96; rows M per batch.
extern "C" int mmsa_convnext_mlp_fused(const unsigned short* Ap, long lda, long strideA, const unsigned short* W1p, long strideW1,
                                       const unsigned short* W2p, long strideW2, const float* b1, const float* b2, const float* gamma,
                                       float* x, long ldx, long strideX, int M, int C, int batch, int max_grid, int fmt, float* clamp_max, hipStream_t stream) {
  MMSA_CHECK_ARG(Ap && W1p && W2p && b1 && b2 && gamma && x && M > 0 && batch > 0, "convnext_mlp_fused: bad args");
  MMSA_CHECK_ARG(C == MF_C, "convnext_mlp_fused: C = %d not supported (%d)", C, MF_C);
  MMSA_CHECK_ARG(fmt == MMSA_FMT_B3 || fmt == MMSA_FMT_F3, "convnext_mlp_fused: planes format %d (bf16 hi/lo or f3)", fmt);
  MMSA_CHECK_ARG((long)batch * 4 * C <= 2048, "convnext_mlp_fused: batch * 4C = %ld > 2048 (bias staging)", (long)batch * 4 * C);
  MMSA_CHECK_ARG(lda >= 2L * C && (lda & 63) == 0 && (strideA & 63) == 0 && (strideW1 & 63) == 0 && (strideW2 & 63) == 0,
                 "convnext_mlp_fused: plane strides must be multiples of 64");
  MMSA_CHECK_ARG(((((uintptr_t)Ap) | ((uintptr_t)W1p) | ((uintptr_t)W2p)) & 127) == 0, "convnext_mlp_fused: planes must be 128-byte aligned");
  MMSA_CHECK_ARG(((((uintptr_t)x) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)gamma)) & 15) == 0 && (ldx & 3) == 0 && (strideX & 3) == 0 && ldx >= C,
                 "convnext_mlp_fused: fp32 pointers must be 16-byte aligned, ldx %% 4 == 0");
  MlpFusedArgs a;
  a.Ap = Ap; a.lda = lda; a.strideA = strideA; a.W1p = W1p; a.strideW1 = strideW1; a.W2p = W2p; a.strideW2 = strideW2;
  a.b1 = b1; a.b2 = b2; a.gamma = gamma; a.x = x; a.ldx = ldx; a.strideX = strideX; a.M = M; a.batch = batch; a.clamp = clamp_max;
  a.tiles_per_batch = cdiv(M, 128);
  a.ntiles = a.tiles_per_batch * batch;
  static MmsaPerDevice per_dev_ = {};
  const int num_cus = mmsa_per_device(per_dev_, [] {
    (void)hipFuncSetAttribute((const void*)mlp_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS);
    (void)hipFuncSetAttribute((const void*)mlp_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS);
  });
  const int cus = (max_grid > 0 && max_grid < num_cus) ? max_grid : num_cus;
  const int grid = a.ntiles < cus ? a.ntiles : cus;
  if (fmt == MMSA_FMT_F3) hipLaunchKernelGGL(mlp_fused_kernel<true>, dim3(grid), dim3(512), MF_LDS, stream, a);
  else hipLaunchKernelGGL(mlp_fused_kernel<false>, dim3(grid), dim3(512), MF_LDS, stream, a);
  MMSA_CHECK_LAUNCH("convnext_mlp_fused");
  return MMSA_OK;
}
