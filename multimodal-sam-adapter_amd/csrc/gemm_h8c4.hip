// h8c GEMM, 4-wave flavour: 128 x 128 tiles, TWO workgroups per CU (round 6; VERDICT r05 "next" item 2a).  Same operand planes, same arithmetic in the same order
// per accumulator (32 fp16 MFMAs, then 16 block-scaled fp8 MFMAs per pair of k-tiles) and the same epilogue include as gemm_h8c.hip's 8-wave kernel: results
// are bit-identical (tests/test_planes_gpu.py); see gemm_v2.hip / gemm_h8c.hip for the reference call sites (IE:488,499; ops/modules/ms_deform_attn.py:103-129).
//
// Why.  A launch of exactly one 256 x 128 tile per CU (proj: 8192 x 1024 x 1024 = 256 tiles; the value / offsets projections on the ViT tokens; the K = 512
// producer of the stream) runs pipeline fill -> k loop -> epilogue once per CU with nothing to overlap: every CU is in its epilogue at the same time, the matrix
// pipes of the whole chip idle.  Round 5's microbenchmark (tools/exp/h8c_2wg.hip, profiles/r05_h8c_2wg_microbench.txt) measured the two-workgroup form 18 %
// faster on exactly that site and 2-17 % slower on the multi-round sites, so this flavour was built for per-SITE dispatch: launches whose 256-row tiling is at
// most one round (mmsa_gemm_v2_launch).  RESULT (profiles/r06_h8c_4wave.txt): bit-identical to the 8-wave kernel on every shape / epilogue tried -- and step-neutral:
// with the site's real epilogue (fp32 rows + operand planes + strip sums + residual) proj takes 58.2 us in this form against 56.8 us; the microbenchmark's gain was
// its lighter epilogue.  The shape dispatch is therefore OFF (gemm_v2.hip MMSA_H8C4_DEFAULT = 0); the kernel stays reachable as `flavour` = 4 (tests, A/B runs).
//
// Structure (the microbenchmark's, on the library's plane layout).  Workgroup = 4 waves (2 x 2, wave tile 64 x 64); LDS = two HI units of 32 KiB (128 A rows +
// 128 W rows x 128 B: the two k-tiles of a 64-k chunk side by side) + ONE LO unit of 16 KiB (64 + 64 row-pair lines) = 80 KiB: two workgroups per CU, one wave
// of each on every SIMD; no ping-pong inside a workgroup -- the other workgroup's wave on the SIMD is what issues beside this one's MFMAs.  Per pair of k-tiles:
// hi fragments (16 ds_read_b128) -> barrier A (HI(p+1), LO(p) landed; unit p & 1 is read) -> HI(p+2) requested into it (8 DMA instructions per wave) -> 32 fp16
// MFMAs -> lo fragments + v_perm_b32 -> barrier B -> LO(p+1) requested (4 instructions) -> 16 fp8 MFMAs.  A tile's stream is drained at its end (the target
// launches have one or two tiles per workgroup; the other workgroup covers the refill): the epilogue stages through HI unit 0.
#include "gemm_v2_shared.h"

typedef __attribute__((ext_vector_type(4))) unsigned h4_u4;

#define H4_H_UNIT 32768
#define H4_L_UNIT 16384
#define H4_LDS_TOTAL (2 * H4_H_UNIT + H4_L_UNIT)

template <bool GEN, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_h8c4_kernel(GemmV2Args a) {
  constexpr bool PP = true;                  // (epilogue include: "the k loop left nothing in flight and every wave past its last LDS read")
  constexpr bool EPI_UNROLL = ACT >= 0;
  constexpr int V2_BM = 128;
  constexpr int V2_NST = 2;                  // (epilogue include: unused on the PP path)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = a.K;
  const int np = K >> 6;                     // k-tile pairs per output tile
  const bool ni4 = true;                     // 128-column tiles only
  const int swid = 64;
  const int G = gridDim.x;
  int rb = blockIdx.x;
  { const int xcd = rb & 7, q = G >> 3, r = G & 7; rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3); }
  if (rb >= a.ntiles) return;

  // ---- DMA lane mapping (gemm_h8c.hip): HI instruction = 8 rows x 128 B, lane -> (row drow, LDS slot lane & 7), piece = slot ^ ((row >> 1) & 7);
  // LO instruction = 8 row-pair lines, lane -> (pair drow, slot), piece = slot ^ f(drow >> 1), f = {0, 3, 2, 1}
  const int drow = lane >> 3;
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 16;
  const int lq = ((lane & 7) ^ ((-(drow >> 1)) & 3)) * 16;
  const int lds_ha = wave * 32 * 128, lds_hw = 16384 + wave * 32 * 128;
  const int lds_la = wave * 2048, lds_lw = 8192 + wave * 2048;
  const unsigned ldaB = (unsigned)(a.lda * 2), ldwB = (unsigned)(a.ldw * 2);   // row-PAIR strides in bytes
  const unsigned K2 = (unsigned)K * 2u, K4 = (unsigned)K * 4u;

  // ---- fragment offsets
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag0 = l15 * 128 + fslot * 16, frag1 = l15 * 128 + (fslot ^ 4) * 16;
  const int fha = (wm * 64) * 128, fhw = 16384 + (wn * 64) * 128;
  const int lo_off = 128 * (l15 >> 1) + 16 * ((((l15 & 1) << 2) | g) ^ ((-(l15 >> 2)) & 3));
  const int fla = (wm * 4) * 1024 + lo_off, flw = 8192 + (wn * 4) * 1024 + lo_off;

  f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j_ = 0; j_ < 4; ++j_) acc[i][j_] = (f32x4){0.f, 0.f, 0.f, 0.f};
  h4_u4 ah0[4], ah1[4], wh0[4], wh1[4];
  mx_v8i opA[4], opW[4];
  int nowait = 0;
  (void)nowait;
  const unsigned psel = 0x07050301u;
#define H4_SB() __builtin_amdgcn_sched_barrier(0)
#define H4_BAR() { H4_SB(); __builtin_amdgcn_s_barrier(); H4_SB(); }
#define H4_PERM(d_, hi_, lo_) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d_) : "v"(hi_), "v"(lo_), "s"(psel))

#pragma unroll 1
  for (int tile = rb; tile < a.ntiles; tile += G) {
    // ---- this tile's operand rows: per-lane source offsets, rows beyond M / N clamped into the matrix (their products are never stored)
    const unsigned char *hA, *hW;
    unsigned oa[4], ow[4], la_[2], lw_[2];
    {
      const int per_b_ = a.nbm * a.nbn;
      const int bz_ = tile / per_b_;
      const int r_ = tile - bz_ * per_b_;
      int tmi_, tni_;
      V2_TILE_MN(r_, tmi_, tni_);
      const int m0_ = tmi_ * 128, n0_ = tni_ * 128;
      hA = reinterpret_cast<const unsigned char*>(a.Ap + (long)bz_ * a.strideA + (long)(m0_ >> 1) * a.lda);
      hW = reinterpret_cast<const unsigned char*>(a.Wp + (long)bz_ * a.strideW + (long)(n0_ >> 1) * a.ldw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned ra = (unsigned)(min(m0_ + wave * 32 + 8 * i + drow, a.M - 1) - m0_);
        const unsigned rw = (unsigned)(min(n0_ + wave * 32 + 8 * i + drow, a.N - 1) - n0_);
        const unsigned pc = (i & 1) ? (unsigned)(dpiece ^ 64) : (unsigned)dpiece;
        oa[i] = (ra >> 1) * ldaB + (ra & 1u) * K2 + pc;
        ow[i] = (rw >> 1) * ldwB + (rw & 1u) * K2 + pc;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned pa = (unsigned)(min((m0_ >> 1) + wave * 16 + 8 * i + drow, (a.M - 1) >> 1) - (m0_ >> 1));
        const unsigned pw = (unsigned)(min((n0_ >> 1) + wave * 16 + 8 * i + drow, (a.N - 1) >> 1) - (n0_ >> 1));
        la_[i] = pa * ldaB + K4 + (unsigned)lq;
        lw_[i] = pw * ldwB + K4 + (unsigned)lq;
      }
    }
#define H4_H_ISSUE(p_)                                                                                          \
  { unsigned char* d_ = smem + H4_H_UNIT * ((p_) & 1); const long ko_ = (long)(p_) * 128;                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                           \
      GLDS16(hA + ko_ + (unsigned long)oa[i_], d_ + lds_ha + 1024 * i_);                                         \
      GLDS16(hW + ko_ + (unsigned long)ow[i_], d_ + lds_hw + 1024 * i_); } }
#define H4_L_ISSUE(p_)                                                                                          \
  { unsigned char* d_ = smem + 2 * H4_H_UNIT; const long ko_ = (long)(p_) * 128;                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                           \
      GLDS16(hA + ko_ + (unsigned long)la_[i_], d_ + lds_la + 1024 * i_);                                        \
      GLDS16(hW + ko_ + (unsigned long)lw_[i_], d_ + lds_lw + 1024 * i_); } }
    // ---- prologue of the tile's stream: HI(0), HI(1), LO(0)
    H4_H_ISSUE(0)
    if (np > 1) H4_H_ISSUE(1)
    H4_L_ISSUE(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H4_BAR()
#pragma unroll 1
    for (int p = 0; p < np; ++p) {
      const unsigned char* hb = smem + H4_H_UNIT * (p & 1);
      const unsigned char* lb = smem + 2 * H4_H_UNIT;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ah0[i] = *reinterpret_cast<const h4_u4*>(hb + fha + i * 2048 + frag0);
        ah1[i] = *reinterpret_cast<const h4_u4*>(hb + fha + i * 2048 + frag1);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wh0[i] = *reinterpret_cast<const h4_u4*>(hb + fhw + i * 2048 + frag0);
        wh1[i] = *reinterpret_cast<const h4_u4*>(hb + fhw + i * 2048 + frag1);
      }
      H4_SB();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // HI(p+1) (requested a pair ago) and LO(p) (half a pair ago)
      H4_BAR()                                             // A: unit p & 1 has been read by every wave; HI(p+1) and LO(p) are visible
      if (p + 2 < np) H4_H_ISSUE(p + 2)
      H4_SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh0[ni]), __builtin_bit_cast(mx_h8, ah0[mi]), acc[ni][mi], 0, 0, 0);
      H4_SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh1[ni]), __builtin_bit_cast(mx_h8, ah1[mi]), acc[ni][mi], 0, 0, 0);
      H4_SB();
      // lo pairs -> fp8 tuples.  A operand: [q(hi) k-tile 0 | q(hi) k-tile 1 | lo 0 | lo 1], W operand: [lo 0 | lo 1 | q(hi) 0 | q(hi) 1] (gemm_h8c.hip)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const h4_u4 la = *reinterpret_cast<const h4_u4*>(lb + fla + i * 1024);
        const h4_u4 lw = *reinterpret_cast<const h4_u4*>(lb + flw + i * 1024);
        int a0, a1, a2, a3, w0, w1, w2, w3;
        H4_PERM(a0, ah0[i][1], ah0[i][0]); H4_PERM(a1, ah0[i][3], ah0[i][2]);
        H4_PERM(a2, ah1[i][1], ah1[i][0]); H4_PERM(a3, ah1[i][3], ah1[i][2]);
        H4_PERM(w0, wh0[i][1], wh0[i][0]); H4_PERM(w1, wh0[i][3], wh0[i][2]);
        H4_PERM(w2, wh1[i][1], wh1[i][0]); H4_PERM(w3, wh1[i][3], wh1[i][2]);
        opA[i] = (mx_v8i){a0, a1, a2, a3, (int)la[0], (int)la[1], (int)la[2], (int)la[3]};
        opW[i] = (mx_v8i){(int)lw[0], (int)lw[1], (int)lw[2], (int)lw[3], w0, w1, w2, w3};
      }
      H4_SB();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      H4_BAR()                                             // B: the LO unit has been read by every wave
      if (p + 1 < np) H4_L_ISSUE(p + 1)
      H4_SB();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, MMSA_H8_MFMA_SCALE, 0, 0x7f7f7f7f);
      H4_SB();
    }
    // ---- tile boundary: nothing is in flight (the last pair requested nothing; its own operands were waited for at its barrier A) and every wave is past its
    // last LDS read (barrier B of the last pair): the epilogue stages through HI unit 0 (4 waves x 16 rows x 68 floats = 17 KiB)
    {
      int lane_o_ = lane;
      asm volatile("" : "+v"(lane_o_));
      const int lane = lane_o_, l15 = lane_o_ & 15, g = lane_o_ >> 4;
#define EPI_STAGING_BASE (smem)
#include "gemm_v2_epilogue.inc"
#undef EPI_STAGING_BASE
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    H4_BAR()   // the staging rows are free again: the next tile's prologue writes HI unit 0
  }
}

// Launch (called by mmsa_gemm_v2_launch in gemm_v2.hip, which builds the argument block for 128-row tiles and picks the grid: at most two workgroups per CU).
int mmsa_gemm_h8c4_dispatch(const GemmV2Args& a, int grid, hipStream_t stream) {
  static MmsaPerDevice per_dev_ = {};   // the kernel's LDS attribute, once per device (common.h)
  (void)mmsa_per_device(per_dev_, [] {
    (void)hipFuncSetAttribute((const void*)gemm_h8c4_kernel<false, ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, H4_LDS_TOTAL);
  });
  hipLaunchKernelGGL((gemm_h8c4_kernel<false, ACT_NONE>), dim3(grid), dim3(256), H4_LDS_TOTAL, stream, a);
  MMSA_CHECK_LAUNCH("gemm_split3(h8c, 4 waves)");
  return MMSA_OK;
}
