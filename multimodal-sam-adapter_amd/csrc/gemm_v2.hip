// split3 GEMM, main-path kernel for gfx950 (activation planes in, any epilogue out).
//
// Same math and epilogue contract as gemm_split3.hip (see there for the reference call sites); this kernel is the
// one the hot shapes run on.  Differences, all driven by what the PMC counters of the first kernel showed
// (MFMA busy 26 %, waves parked 45 % of their life at s_waitcnt / s_barrier):
//   * 256 x 128 block tile, 8 waves (4 x 2), wave tile 64 x 64: one workgroup per CU, 2 waves per SIMD;
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write issue slots);
//   * 3-stage LDS ring (3 x 48 KiB), DMA for k-tile t+2 is issued while t is computed, a COUNTED s_waitcnt
//     vmcnt (the DMA instructions of tile t+1 stay in flight) and ONE raw s_barrier per k-tile;
//   * PERSISTENT workgroups (grid = min(#tiles, #CUs)): each walks its output tiles and the (tile, k-tile) pairs
//     form one continuous DMA stream, so the first k-tiles of the next output tile are already landing while the
//     epilogue of the current one runs (no per-tile fill/drain, no workgroup relaunch);
//   * operands are stored as INTERLEAVED hi/lo planes (common.h: one k-block of one row = one 128-byte line), so a
//     DMA instruction fetches 8 rows x 128 B = full lines (the two-array layout issued 64-byte half-line requests
//     and ran at ~50 % of the L2 request rate: TCC_REQ ~ 2x bytes/128);
//   * LDS image row-major, 128 B per row = 8 slots of 16 B (4 hi chunks | 4 lo chunks), slot index XORed with
//     (row>>1)&7: every ds_read_b128 fragment read (hi and lo) is bank-conflict free; a DMA instruction writes
//     1 KiB linearly, so the permutation is applied on the per-lane SOURCE address;
//   * compact epilogue: a ROLLED loop over the four 16-row sub-tiles (accumulator columns rotated by register moves),
//     one activation branch per 4 values: it must stay well inside the 64 KiB instruction cache (see the epilogue);
//   * XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) walk neighbouring (m-tile, n-tile) pairs so
//     activation rows and weight panels are re-read from that XCD's L2 (speed only, never correctness).
#include "common.h"
#include "gemm_v2_shared.h"
// NW = waves per workgroup.  8: the 256 x 128 tile, 3-slot ring, one workgroup per CU (ping-pong main loop).  4: a 128 x 128 tile
// (wave tile 64 x 64 as before), 2-slot ring of 32 KiB stages, TWO workgroups per CU: the epilogue of one workgroup (VALU + stores,
// matrix pipe idle) runs under the k-loop of the other.  For shapes whose epilogue is a large share of a tile's life (few
// k-tiles: K <= 512, GELU / planes epilogues) that overlap is worth more than the deeper pipeline of the big tile.
// FMT = operand format of A and W (common.h): MMSA_FMT_B3 = bf16 hi/lo planes, three bf16 MFMAs per k-tile and output tile;
// MMSA_FMT_H8 = fp16 hi + e5m2 cross-term bytes: ONE fp16 MFMA per k-tile plus ONE block-scaled fp8 MFMA (K = 128: both cross terms
// of two k-tiles) per PAIR of k-tiles -- 2/3 of the matrix-pipe time of the bf16 scheme at the same operand bytes; the k loop is then
// unrolled by two (K % 64 == 0) so that the fp8 operand tuples are assembled in place by the fragment reads of the two k-tiles.
template <bool GEN, int ACT, bool PP, int NW, int FMT>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void gemm_v2_kernel(GemmV2Args a) {
  static_assert(NW == 8 || (NW == 4 && !PP), "4-wave workgroups run the in-phase main loop");
  static_assert(FMT == MMSA_FMT_B3 || (PP && NW == 8), "the h8 and f3 operand formats run on the ping-pong kernel");
  constexpr bool EPI_UNROLL = ACT >= 0;
  constexpr int V2_BM = NW * 32;
  constexpr int V2_A_BYTES = V2_BM * 128;
  constexpr int V2_STAGE = V2_A_BYTES + V2_W_BYTES;
  constexpr int V2_NST = NW == 8 ? 3 : 2;
  constexpr int WROWS = V2_BN / NW;          // weight rows a wave stages per k-tile: 16 (two DMA instructions) or 32 (four)
  constexpr int NDMA = 4 + WROWS / 8;        // DMA instructions per wave and k-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int grp = wave >> 2;   // PP: wave w runs on SIMD w % 4, so every SIMD hosts one wave of each group
  const int l15 = lane & 15, g = lane >> 4;
  const int K = a.K;
  const int nk = K / V2_BK;

  // XCD-aware logical id: blocks with equal blockIdx % 8 (same XCD under round-robin placement) get
  // consecutive logical ids, hence neighbouring tiles.  Bijective for any grid size.
  const int G = gridDim.x;
  int rb = blockIdx.x;
  {
    const int xcd = rb & 7, q = G >> 3, r = G & 7;
    rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3);
  }
  const int my_tiles = (a.ntiles - rb + G - 1) / G;   // tiles rb, rb+G, ...
  if (my_tiles <= 0) return;
  V2_SLACK_STAGGER(a, rb, G)
  CLK_SAMPLE(0)
  const int total = my_tiles * nk;

  // ---- DMA: one instruction = 8 rows x 128 B; lane -> (row = lane>>3, slot = lane&7); the 16-byte piece fetched
  //      for LDS slot s of row r is piece = s ^ ((r>>1)&7)   (piece 0-3: hi chunks, 4-7: lo chunks)
  const int drow = lane >> 3;
  // rows 8i + drow of a 16-row tile: key = (row_in_16 >> 1) & 7 = (drow >> 1) + 4*(i & 1)
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 8;         // even 8-row groups; odd groups use dpiece ^ 32
  const int lds_a = wave * 32 * 128;   // this wave's 32 rows of the A image (4 instructions x 8 rows)
  const int lds_w = wave * WROWS * 128;   // this wave's rows of the W image (2 or 4 instructions)
  // DMA source of piece i = wave-uniform base of the output tile's first row (SGPR pair) + a 32-bit per-lane byte offset: the
  // `global_load_lds ... v_off, s[base]` form -- one scalar add per k-tile and operand instead of a 64-bit vector add per piece.
  const unsigned char *gA, *gW;
  unsigned oa0, oa1, oa2, oa3, ow0, ow1, ow2 = 0, ow3 = 0;
#define SA(i_, ko_) reinterpret_cast<const unsigned short*>(gA + (long)(ko_) * 2 + (unsigned long)oa##i_)
#define SW(i_, ko_) reinterpret_cast<const unsigned short*>(gW + (long)(ko_) * 2 + (unsigned long)ow##i_)

#define SET_TILE_SRC(tile_)                                                      \
  do {                                                                           \
    const int t_ = (tile_);                                                      \
    const int per_b_ = a.nbm * a.nbn;                                            \
    const int bz_ = t_ / per_b_;                                                 \
    const int r_ = t_ - bz_ * per_b_;                                            \
    int tmi_, tni_;                                                              \
    V2_TILE_MN(r_, tmi_, tni_);                                                  \
    const int m0_ = tmi_ * V2_BM, n0_ = tni_ * a.bn;                             \
    const int ab_ = m0_ + wave * 32 + drow, wb_ = n0_ + wave * WROWS + drow;     \
    gA = reinterpret_cast<const unsigned char*>(a.Ap + (long)bz_ * a.strideA + (long)m0_ * a.lda);  \
    gW = reinterpret_cast<const unsigned char*>(a.Wp + (long)bz_ * a.strideW + (long)n0_ * a.ldw);  \
    oa0 = (unsigned)((min(ab_, a.M - 1) - m0_) * (int)a.lda + dpiece) * 2u;           \
    oa1 = (unsigned)((min(ab_ + 8, a.M - 1) - m0_) * (int)a.lda + (dpiece ^ 32)) * 2u;  \
    oa2 = (unsigned)((min(ab_ + 16, a.M - 1) - m0_) * (int)a.lda + dpiece) * 2u;      \
    oa3 = (unsigned)((min(ab_ + 24, a.M - 1) - m0_) * (int)a.lda + (dpiece ^ 32)) * 2u; \
    ow0 = (unsigned)((min(wb_, a.N - 1) - n0_) * (int)a.ldw + dpiece) * 2u;           \
    ow1 = (unsigned)((min(wb_ + 8, a.N - 1) - n0_) * (int)a.ldw + (dpiece ^ 32)) * 2u;  \
    if constexpr (NW == 4) {                                                     \
      ow2 = (unsigned)((min(wb_ + 16, a.N - 1) - n0_) * (int)a.ldw + dpiece) * 2u;    \
      ow3 = (unsigned)((min(wb_ + 24, a.N - 1) - n0_) * (int)a.ldw + (dpiece ^ 32)) * 2u; \
    }                                                                            \
    if (V2_DBG(a) == 4 || V2_DBG(a) == 5) { oa1 = oa2 = oa3 = oa0; gW = gA; ow0 = ow1 = oa0; }   /* timing experiment: L1-resident operand stream */ \
  } while (0)

#define ISSUE_DMA(kt_, st_)                                                       \
  do {                                                                            \
    unsigned char* sb_ = smem + (st_) * V2_STAGE;                                 \
    const int ko_ = V2_DBG(a) == 3 ? 0 : (kt_) * 64;   /* debug 3: timing experiment, re-read k-tile 0 */ \
    GLDS16(SA(0, ko_), sb_ + lds_a);                                               \
    GLDS16(SA(1, ko_), sb_ + lds_a + 1024);                                        \
    GLDS16(SA(2, ko_), sb_ + lds_a + 2048);                                        \
    GLDS16(SA(3, ko_), sb_ + lds_a + 3072);                                        \
    GLDS16(SW(0, ko_), sb_ + V2_A_BYTES + lds_w);                                  \
    GLDS16(SW(1, ko_), sb_ + V2_A_BYTES + lds_w + 1024);                           \
    if constexpr (NW == 4) {                                                      \
      GLDS16(SW(2, ko_), sb_ + V2_A_BYTES + lds_w + 2048);                         \
      GLDS16(SW(3, ko_), sb_ + V2_A_BYTES + lds_w + 3072);                         \
    }                                                                             \
  } while (0)

  f32x4 acc[4][4];  // [ni][mi]
  mx_v8i opA[4] = {}, opW[4] = {};   // h8: the fp8 operands of a k-tile pair (unused, and optimised away, for bf16 planes)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = l15 within a 16-row tile, hi chunk g at slot g ^ ((row>>1)&7), lo at slot ^ 4
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag_hi = l15 * 128 + fslot * 16;
  const int frag_lo = l15 * 128 + (fslot ^ 4) * 16;
  const int fa = (wm * 64) * 128;                    // activation rows of this wave, + mi*2048
  const bool ni4 = a.bn == V2_BN;                    // wave-uniform: four n-tiles per wave (128-column tiles) or three (96)
  const int swid = ni4 ? 64 : 48;                    // columns of this wave's strip
  const int fw = V2_A_BYTES + (wn * swid) * 128;     // weight rows of this wave, + ni*2048

  // ---- prefetch cursor (runs 2 iterations ahead of the compute cursor)
  int pf_tile = rb, pf_kt = 0, pf_st = 0, pf_j = 0;
  SET_TILE_SRC(pf_tile);
#define PREFETCH_NEXT()                                     \
  do {                                                      \
    ISSUE_DMA(pf_kt, pf_st);                                \
    pf_st = pf_st == V2_NST - 1 ? 0 : pf_st + 1;            \
    ++pf_j;                                                 \
    if (++pf_kt == nk) {                                    \
      pf_kt = 0;                                            \
      pf_tile += G;                                         \
      if (pf_j < total) SET_TILE_SRC(pf_tile);              \
    }                                                       \
  } while (0)
  PREFETCH_NEXT();
  if (V2_NST == 3 && total > 1) PREFETCH_NEXT();   // the 3-slot ring runs two k-tiles ahead, the 2-slot ring one

  int st = 0, tile = rb, nowait = 0, j = 0;
  if constexpr (PP) {   // k-tile 0 must be visible before the first read phase
    if (total > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // One k-tile.  The DMAs of iteration j must have landed; those of j+1 (6 instructions, issued later) may stay in
  // flight.  vmcnt retires in order and counts stores too, so a wait issued after an epilogue would also wait for that
  // epilogue's stores; instead the epilogue first drains the (older) DMAs of j+1 and j+2 and the next two iterations skip
  // the wait.  The 6 DMA instructions of iteration j+2 (ring slot (j+2)%3, last read in iteration j-1: free after the
  // barrier) are issued BETWEEN the MFMA chunks, two per chunk: an LDS-DMA costs ~100 issue cycles (M0 write, address
  // arithmetic, the instruction) which disappear under the 16-cycle passes of the MFMAs already queued.
// MXPAR (a literal 0 / 1 defined around every expansion) = parity of the k-tile inside its pair, h8 operands only.
#define MX_SET(dst_, src_, par_)                                                                            \
  { const uint4 u_ = __builtin_bit_cast(uint4, src_);                                                       \
    if ((par_) == 0) { dst_[0] = (int)u_.x; dst_[1] = (int)u_.y; dst_[2] = (int)u_.z; dst_[3] = (int)u_.w; }  \
    else { dst_[4] = (int)u_.x; dst_[5] = (int)u_.y; dst_[6] = (int)u_.z; dst_[7] = (int)u_.w; } }
// h8, second k-tile of a pair: the 16 block-scaled fp8 MFMAs (cross terms of both k-tiles) go FIRST, the fp16 MFMAs last.  MFMAs are
// queued: a wave reaches the barrier that ends its matrix phase with its last MFMAs still waiting for the pipe, and its next read
// phase overwrites the fp8 operand tuples -- every fragment read of that phase then stalls until the queued fp8 MFMAs have read
// them (interval stamps: the read phase after an fp8 burst took 1150-1400 cycles, the other one 600).  With the fp16 MFMAs at the
// tail the registers still in use are the hi fragments of the finished k-tile, which the next read phase does not touch.
// The matrix pipe's share of a k-tile pair is then 16 fp16 MFMAs (256 cycles) in the first step and 16 fp8 + 16 fp16 (768) in the second,
// beside read phases of ~700 cycles on the partner group (the L2 -> LDS operand stream): the second step's phase is matrix-bound, the
// first one's read-bound, 3350 cycles per pair where 4 x 700 would do.  (Spreading the fp8 MFMAs over both steps -- round 3's the fp8-split experiment --
// was built, measured slower (profiles/r03_fp8_split_ab.txt) and removed in round 4.)
#define MFMA_FP8_ALL(NI4_, DEFER_)                                                                          \
  if constexpr (FMT == MMSA_FMT_H8) {                                                                       \
    if (MXPAR && V2_FP8_FIRST && !V2_EXP_NO_FP8) {                                                          \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                      \
        if (ni < 3 || (NI4_)) {                                                                             \
          _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
              acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, MMSA_H8_MFMA_SCALE, 0, 0x7f7f7f7f); \
        }                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
  }
#define MFMA_CHUNK(ni)                                                                                      \
  if constexpr (FMT == MMSA_FMT_H8) {                                                                       \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                      \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh[ni]), __builtin_bit_cast(mx_h8, ah[mi]), acc[ni][mi], 0, 0, 0); \
      if (MXPAR && !V2_FP8_FIRST && !V2_EXP_NO_FP8) acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, MMSA_H8_MFMA_SCALE, 0, 0x7f7f7f7f); \
    }                                                                                                       \
  } else {                                                                                                  \
    if (V2_SETPRIO) __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                                      \
      if constexpr (FMT == MMSA_FMT_F3) {   /* fp16 hi/lo pairs: the same three products on the fp16 MFMA */                 \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wl[ni]), __builtin_bit_cast(mx_h8, ah[mi]), acc[ni][mi], 0, 0, 0); \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh[ni]), __builtin_bit_cast(mx_h8, al[mi]), acc[ni][mi], 0, 0, 0); \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh[ni]), __builtin_bit_cast(mx_h8, ah[mi]), acc[ni][mi], 0, 0, 0); \
      } else {                                                                                              \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);          \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);          \
      acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);          \
      }                                                                                                     \
    }                                                                                                       \
    if (V2_SETPRIO) __builtin_amdgcn_s_setprio(0);                                                          \
  }
#define K_STEP()                                                                                            \
  {                                                                                                         \
    if (nowait > 0) --nowait;                                                                               \
    else if (V2_NST == 3 && j + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    __builtin_amdgcn_s_barrier();                                                                           \
    const bool do_pf = pf_j < total;                                                                        \
    unsigned char* pfb = smem + pf_st * V2_STAGE;                                                           \
    const int pko = V2_DBG(a) == 3 ? 0 : pf_kt * 64;                                                          \
    const unsigned char* base = smem + st * V2_STAGE;                                                       \
    bf16x8 ah[4], al[4], wh[4], wl[4];                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      ah[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_hi);                             \
      al[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_lo);                             \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      if (i < 3 || ni4) {                                                                                   \
        wh[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_hi);                           \
        wl[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_lo);                           \
      }                                                                                                     \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    MFMA_CHUNK(0)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (do_pf) { GLDS16(SA(0, pko), pfb + lds_a); GLDS16(SA(1, pko), pfb + lds_a + 1024); }                   \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    MFMA_CHUNK(1)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (do_pf) { GLDS16(SA(2, pko), pfb + lds_a + 2048); GLDS16(SA(3, pko), pfb + lds_a + 3072); }            \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    MFMA_CHUNK(2)                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (do_pf) { GLDS16(SW(0, pko), pfb + V2_A_BYTES + lds_w); GLDS16(SW(1, pko), pfb + V2_A_BYTES + lds_w + 1024); } \
    if constexpr (NW == 4) { if (do_pf) { GLDS16(SW(2, pko), pfb + V2_A_BYTES + lds_w + 2048); GLDS16(SW(3, pko), pfb + V2_A_BYTES + lds_w + 3072); } } \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (ni4) { MFMA_CHUNK(3) }                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (do_pf) {   /* advance the prefetch cursor (source pointers of the next output tile when the k loop wraps) */ \
      pf_st = pf_st == V2_NST - 1 ? 0 : pf_st + 1;                                                          \
      ++pf_j;                                                                                               \
      if (++pf_kt == nk) {                                                                                  \
        pf_kt = 0;                                                                                          \
        pf_tile += G;                                                                                       \
        if (pf_j < total) SET_TILE_SRC(pf_tile);                                                            \
      }                                                                                                     \
    }                                                                                                       \
    st = st == V2_NST - 1 ? 0 : st + 1;                                                                     \
    ++j;                                                                                                    \
  }

  // Ping-pong k-tile (PP).  Barriers are numbered b1, b2, ... per output tile; group 1 passes one extra barrier before its
  // first k-tile and group 0 one after its last, so between b(2j) and b(2j+1) group 0 READS k-tile j while group 1 runs the
  // MFMAs of j-1, and between b(2j+1) and b(2j+2) group 0 runs the MFMAs of j while group 1 reads j.
  //   * a read phase = 16 ds_read_b128 + this wave's 6 DMA instructions of k-tile j+2 (ring slot (j+2)%3 = (j-1)%3: both
  //     groups retired their reads of it with lgkmcnt(0) before b(2j)) + lgkmcnt(0);
  //   * k-tile j+1 must have landed for ALL waves before b(2j+2) (group 0 reads it right after): group 0 waits at the end of
  //     its MFMA phase, group 1 at the end of its read phase, both with the 6 DMAs of j+2 left in flight;
  //   * last k-tile of an output tile: everybody drains (vmcnt 0) before b(2nk+1); after it the epilogue owns slot st_cur.
#ifdef V2_STAMP
#define STAMP_DECL() unsigned long long tt[9]; const bool stamp_on = blockIdx.x == 0 && tdone == 0 && kt >= 8 && kt < 12;
#define STAMP_STORE() if (stamp_on && lane == 0) { _Pragma("unroll") for (int q_ = 0; q_ < 9; ++q_) g_v2_stamps[(wave * 4 + (kt - 8)) * 10 + q_] = tt[q_]; }
#else
#define STAMP_DECL()
#define STAMP_STORE()
#endif
#define V2_PP_NR 6   // all 6 DMA pieces of a wave and k-tile are issued in the READ phase: moving 2..6 of them between the MFMA
                     // chunks (where a piece costs fewer issue cycles) measured 5-10 % slower (same-box A/B) -- the MFMA phase must stay pure
#define V2_STR_(x) #x
#define V2_STR(x) V2_STR_(x)
#define PP_WAIT(cnt_)                                                                                       \
  {                                                                                                         \
    if (nowait > 0) --nowait;                                                                               \
    else if (j + 2 < total) asm volatile("s_waitcnt vmcnt(" V2_STR(cnt_) ")" ::: "memory");                  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
  }
#define PP_PIECE(i_)                                                                                        \
  if (do_pf) {                                                                                              \
    if ((i_) == 0) GLDS16(SA(0, pko), pfb + lds_a);                                                          \
    if ((i_) == 1) GLDS16(SA(1, pko), pfb + lds_a + 1024);                                                   \
    if ((i_) == 2) GLDS16(SA(2, pko), pfb + lds_a + 2048);                                                   \
    if ((i_) == 3) GLDS16(SA(3, pko), pfb + lds_a + 3072);                                                   \
    if ((i_) == 4) GLDS16(SW(0, pko), pfb + V2_A_BYTES + lds_w);                                             \
    if ((i_) == 5) GLDS16(SW(1, pko), pfb + V2_A_BYTES + lds_w + 1024);                                      \
  }
#define MX_FILL() if constexpr (FMT == MMSA_FMT_H8) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { MX_SET(opA[i], al[i], MXPAR) MX_SET(opW[i], wl[i], MXPAR) } }
// k-loop ablation build (tools/build_variant.sh -DV2_KABL, timing only, results are garbage): MMSA_GEMM_DEBUG = 64 + a bit mask of what
// to leave out -- 1 the MFMAs, 2 the fragment reads, 4 the LDS-DMA, 8 the barriers inside the k loop; all without the epilogue.
#ifdef V2_KABL
#define KABL(x_) (V2_DBG(a) >= 64 && ((V2_DBG(a) >> ((x_) - 6)) & 1))
#define KABL_INIT() _Pragma("unroll") for (int i = 0; i < 4; ++i) { ah[i] = al[i] = wh[i] = wl[i] = __builtin_bit_cast(bf16x8, make_uint4(lane + i, 0x3c003c00u, lane, 0x3c003c00u)); }
#define KABL_UNDEF() _Pragma("unroll") for (int i = 0; i < 4; ++i) { asm volatile("" : "=v"(ah[i]), "=v"(al[i]), "=v"(wh[i]), "=v"(wl[i])); }   /* fragments = whatever the registers hold */
#else
#define KABL(x_) false
#define KABL_INIT()
#define KABL_UNDEF()
#endif
#define K_STEP_PP()                                                                                         \
  {                                                                                                         \
    const bool do_pf = pf_j < total && !KABL(8);                                                            \
    const bool last_k = kt == nk - 1;                                                                       \
    STAMP_DECL()                                                                                            \
    STAMP(0)                                                                                                \
    unsigned char* pfb = smem + pf_st * V2_STAGE;                                                           \
    const int pko = V2_DBG(a) >= 3 && V2_DBG(a) <= 5 ? 0 : pf_kt * 64;                                          \
    const unsigned char* base = smem + st * V2_STAGE;                                                       \
    bf16x8 ah[4], al[4], wh[4], wl[4];                                                                      \
    KABL_INIT()                                                                                             \
    if (!KABL(7)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      ah[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_hi);                             \
      al[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_lo);                             \
    }                                                                                                       \
    if (!KABL(7)) _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
      if (i < 3 || ni4) {                                                                                   \
        wh[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_hi);                           \
        wl[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_lo);                           \
      }                                                                                                     \
    }                                                                                                       \
    MX_FILL()                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    STAMP(1)                                                                                                \
    _Pragma("unroll") for (int pi_ = 0; pi_ < V2_PP_NR; ++pi_) PP_PIECE(pi_)                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    STAMP(2)                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    STAMP(3)                                                                                                \
    if (grp && !last_k) PP_WAIT(V2_PP_NR)                                                                   \
    STAMP(4)                                                                                                \
    if (!KABL(9)) __builtin_amdgcn_s_barrier();                                                             \
    STAMP(5)                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (!KABL(6)) {                                                                                         \
    MFMA_FP8_ALL(ni4, !last_k)                                                                              \
    MFMA_CHUNK(0)                                                                                           \
    MFMA_CHUNK(1)                                                                                           \
    MFMA_CHUNK(2)                                                                                           \
    if (ni4) { MFMA_CHUNK(3) }                                                                              \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (do_pf) {                                                                                            \
      pf_st = pf_st == 2 ? 0 : pf_st + 1;                                                                   \
      ++pf_j;                                                                                               \
      if (++pf_kt == nk) {                                                                                  \
        pf_kt = 0;                                                                                          \
        pf_tile += G;                                                                                       \
        if (pf_j < total) SET_TILE_SRC(pf_tile);                                                            \
      }                                                                                                     \
    }                                                                                                       \
    STAMP(6)                                                                                                \
    if (last_k) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                            \
    else if (!grp) PP_WAIT(6)                                                                               \
    STAMP(7)                                                                                                \
    if (!KABL(9)) __builtin_amdgcn_s_barrier();                                                             \
    STAMP(8)                                                                                                \
    STAMP_STORE()                                                                                           \
    st = st == 2 ? 0 : st + 1;                                                                              \
    ++j;                                                                                                    \
  }

// Steady-state k-tile of the ping-pong loop: the same step with every case distinction of K_STEP_PP resolved -- the prefetch cursor
// stays inside the current output tile (kt + 2 < nk), no wait is skipped (kt >= 2) and no drain is due, so the step is straight-line
// code between its two barriers.  K_STEP_PP spends ~10 scalar branches per step on those cases, and a wave's taken branches sit on
// the critical path between two barriers (the partner group cannot start its phase before this one arrives).  Both groups execute
// both counted waits: the one a group does not need is already satisfied.  NI4_ = four n-tiles per wave (literal).
#define K_STEP_PP_FAST(NI4_)                                                                                \
  {                                                                                                         \
    unsigned char* pfb = smem + pf_st * V2_STAGE;                                                           \
    const int pko = pf_kt * 64;                                                                             \
    const unsigned char* base = smem + st * V2_STAGE;                                                       \
    bf16x8 ah[4], al[4], wh[4], wl[4];                                                                      \
    if (KABL(7)) { KABL_UNDEF() } else {                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      ah[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_hi);                             \
      al[i] = *reinterpret_cast<const bf16x8*>(base + fa + i * 2048 + frag_lo);                             \
    }                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      if (i < 3 || NI4_) {                                                                                  \
        wh[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_hi);                           \
        wl[i] = *reinterpret_cast<const bf16x8*>(base + fw + i * 2048 + frag_lo);                           \
      }                                                                                                     \
    }                                                                                                       \
    }                                                                                                       \
    MX_FILL()                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (!KABL(8)) {                                                                                         \
    GLDS16(SA(0, pko), pfb + lds_a);                                                                        \
    GLDS16(SA(1, pko), pfb + lds_a + 1024);                                                                 \
    GLDS16(SA(2, pko), pfb + lds_a + 2048);                                                                 \
    GLDS16(SA(3, pko), pfb + lds_a + 3072);                                                                 \
    GLDS16(SW(0, pko), pfb + V2_A_BYTES + lds_w);                                                           \
    GLDS16(SW(1, pko), pfb + V2_A_BYTES + lds_w + 1024);                                                    \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                        \
    __builtin_amdgcn_s_barrier();                                                                           \
    ISTAMP(0)                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (!KABL(6)) {                                                                                         \
    MFMA_FP8_ALL(NI4_, true)                                                                                \
    MFMA_CHUNK(0)                                                                                           \
    MFMA_CHUNK(1)                                                                                           \
    MFMA_CHUNK(2)                                                                                           \
    if (NI4_) { MFMA_CHUNK(3) }                                                                             \
    }                                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    pf_st = pf_st == 2 ? 0 : pf_st + 1;                                                                     \
    ++pf_j;                                                                                                 \
    ++pf_kt;                                                                                                \
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                        \
    __builtin_amdgcn_s_barrier();                                                                           \
    ISTAMP(1)                                                                                               \
    st = st == 2 ? 0 : st + 1;                                                                              \
    ++j;                                                                                                    \
  }
// interval stamps (-DV2_STAMP): the shader clock right after each barrier of the straight-line steps kt = 8..15 of workgroup 0's first
// tile, waves 0 (group 0) and 4 (group 1), stored by lane 0
#ifdef V2_STAMP
#ifndef V2_ISTAMP_KT0
#define V2_ISTAMP_KT0 8
#endif
#define ISTAMP(h_) if (blockIdx.x == 0 && tdone == 0 && kt >= V2_ISTAMP_KT0 && kt < V2_ISTAMP_KT0 + 8 && (wave & 3) == 0 && lane == 0) g_v2_stamps[grp * 16 + (kt + MXPAR - V2_ISTAMP_KT0) * 2 + (h_)] = __builtin_readcyclecounter();
#else
#define ISTAMP(h_)
#endif

  for (int tdone = 0; tdone < my_tiles; ++tdone) {
    if constexpr (PP) {
      if (grp) __builtin_amdgcn_s_barrier();
      // k-tiles 0, 1 (a wait may be skipped after an epilogue) and nk-2, nk-1 (the prefetch cursor moves to the next output tile, the
      // last one drains) run the general step, everything between them the straight-line one.
      const bool fast_ok = V2_FAST_STEPS && nk >= 6 && (V2_DBG(a) < 3 || V2_DBG(a) == 10 || V2_DBG(a) >= 64) && (FMT != MMSA_FMT_H8 || ni4);   // h8: a second copy of the loop for 96-column tiles costs registers (spills)
      const int kt_a = fast_ok ? 2 : nk, kt_b = fast_ok ? nk - 2 : nk;
      if constexpr (FMT == MMSA_FMT_H8) {   // nk is even (checked by the launcher): the fp8 operand tuples are filled by a PAIR of k-tiles
#pragma unroll 1
        for (int kt = 0; kt < kt_a; ++kt) {
#define MXPAR 0
          K_STEP_PP()
#undef MXPAR
          ++kt;
#define MXPAR 1
          K_STEP_PP()
#undef MXPAR
        }
#pragma unroll 1
        for (int kt = kt_a; kt < kt_b; kt += 2) {
#define MXPAR 0
          K_STEP_PP_FAST(true)
#undef MXPAR
#define MXPAR 1
          K_STEP_PP_FAST(true)
#undef MXPAR
        }
        if (fast_ok) {   // the straight-line steps left the prefetch cursor at k-tile nk of this output tile: move it to the next one
          pf_kt = 0;
          pf_tile += G;
          if (pf_j < total) SET_TILE_SRC(pf_tile);
        }
#pragma unroll 1
        for (int kt = kt_b; kt < nk; ++kt) {
#define MXPAR 0
          K_STEP_PP()
#undef MXPAR
          ++kt;
#define MXPAR 1
          K_STEP_PP()
#undef MXPAR
        }
      } else {
#define MXPAR 0
#pragma unroll 1
        for (int kt = 0; kt < kt_a; ++kt) K_STEP_PP()
        if (ni4) {
#pragma unroll 1
          for (int kt = kt_a; kt < kt_b; ++kt) K_STEP_PP_FAST(true)
        } else {
#pragma unroll 1
          for (int kt = kt_a; kt < kt_b; ++kt) K_STEP_PP_FAST(false)
        }
        if (fast_ok) {   // the straight-line steps left the prefetch cursor at k-tile nk of this output tile: move it to the next one
          pf_kt = 0;
          pf_tile += G;
          if (pf_j < total) SET_TILE_SRC(pf_tile);
        }
#pragma unroll 1
        for (int kt = kt_b; kt < nk; ++kt) K_STEP_PP()
#undef MXPAR
      }
      if (!grp) __builtin_amdgcn_s_barrier();
    } else {
#define MXPAR 0
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) K_STEP()
#undef MXPAR
    }
    const int st_cur = st == 0 ? V2_NST - 1 : st - 1;   // ring slot of the k-tile just consumed: free until the next DMA into it (issued after the next barrier)

    if (V2_DBG(a) == 2 || V2_DBG(a) == 5 || V2_DBG(a) >= 64) { tile += G; continue; }
    {
      // the epilogue sees the lane id through an opaque copy (as in gemm_h8c.hip): what it derives from it -- row / column indices, 64-bit
      // addresses -- is then computed per tile instead of being hoisted above the tile loop, where those values were live across the k loops
      // and pushed loop invariants into scratch (round 4 ISA: 143-184 scratch instructions in the h8-line flavours; a scratch reload in
      // front of an LDS-DMA instruction is an s_waitcnt vmcnt(0), i.e. a drain of the prefetch stream)
      int lane_o_ = lane;
      asm volatile("" : "+v"(lane_o_));
      const int lane = lane_o_, l15 = lane_o_ & 15, g = lane_o_ >> 4;
#define EPI_STAGING_BASE (smem + st_cur * V2_STAGE)
#include "gemm_v2_epilogue.inc"
#undef EPI_STAGING_BASE
    }
    if constexpr (PP) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // slot st_cur (epilogue staging of every wave) is free again: group 0 DMAs into it next
    }
    tile += G;
  }
  CLK_SAMPLE(2)
#undef K_STEP
#undef K_STEP_PP
#undef K_STEP_PP_FAST
#undef PP_WAIT
#undef PP_PIECE
#undef MFMA_CHUNK
#undef MFMA_FP8_ALL
}

int mmsa_gemm_h8c_dispatch(const GemmV2Args& a, int grid, bool gen, int act, hipStream_t stream);   // gemm_h8c.hip
int mmsa_gemm_stream_try(const unsigned short* Ap, long lda, const unsigned short* Wp, const float* bias, const float* colscale, const float* resid, long ldr, float beta,
                         float* C, long ldc, unsigned short* Cp, long ldcp, int M, int N, int K, int batch, int act, float alpha, int out_mode, int resid_mod,
                         int fmt, int cp_fmt, int max_grid, const float* rs_out, const float* rn_mr, int flavour, float* clamp_max, hipStream_t stream);   // gemm_stream.hip
int mmsa_gemm_h8c4_dispatch(const GemmV2Args& a, int grid, hipStream_t stream);                      // gemm_h8c4.hip (4 waves, 128 x 128 tiles, two workgroups per CU)

// Internal launcher, called by mmsa_gemm_split3 (gemm_split3.hip) after argument validation when A comes as planes.
// fmt = format of the A and W planes, cp_fmt = format of the planes output (common.h).
int mmsa_gemm_v2_launch(const unsigned short* Ap, long lda, long strideA,
                        const unsigned short* Wp, long strideW,
                        const float* bias, long strideBias, const float* colscale,
                        const float* resid, long ldr, long strideR, int resid_mod, float beta,
                        float* C, long ldc, long strideC,
                        unsigned short* Cp, long ldcp, long strideCp,
                        int M, int N, int K, int batch, int act, float alpha,
                        int out_mode, int ps_H, int ps_W, int ps_C, int fmt, int cp_fmt, int max_grid, hipStream_t stream,
                        float* rs_out, const float* rn_mr, const float* rn_cs, int flavour, float* clamp_max) {
  // the skinny launches of the neck (M in the tens of thousands, N and K <= 192): a streaming kernel with the whole weight matrix resident in LDS (gemm_stream.hip)
  {
    const int rc_ = mmsa_gemm_stream_try(Ap, lda, Wp, bias, colscale, resid, ldr, beta, C, ldc, Cp, ldcp, M, N, K, batch, act, alpha, out_mode, resid_mod, fmt, cp_fmt,
                                         max_grid, rs_out, rn_mr, flavour, clamp_max, stream);
    if (rc_ != 0) return rc_ < 0 ? rc_ : MMSA_OK;
  }
  GemmV2Args a;
  a.clamp_max = clamp_max;
  // LayerNorm fold (mmsa_gemm_next_extras): both forms run on the unrolled fast epilogue of 128-column tiles with whole 64-column strips
  MMSA_CHECK_ARG(!rs_out || (C && out_mode == 0 && resid_mod <= 0 && act == ACT_NONE && (N & 63) == 0 && (ldc & 3) == 0 && (!resid || (ldr & 3) == 0) && (!Cp || (ldcp & 3) == 0)),
                 "gemm(v2): row statistics need a plain fp32 output, no activation, N %% 64 == 0 (N=%d)", N);
  // row-normalising epilogue: a planes-only output (any M: ragged tiles run the staged form), or -- round 5, the adapter tokens' LayerNorm folded into value / offset
  // projections and fc1 -- an fp32-only output of whole tiles (M %% 256 == 0: only the register-resident "rows" epilogue has that form)
  MMSA_CHECK_ARG(!rn_mr || (rn_cs && !resid && out_mode == 0 && resid_mod <= 0 && (N & 127) == 0 && ((((uintptr_t)bias) | ((uintptr_t)colscale) | ((uintptr_t)rn_cs)) & 15) == 0 && (strideBias & 3) == 0 &&
                            ((Cp && !C && (act == ACT_NONE || act == ACT_GELU || act == ACT_RELU)) ||
                             (C && !Cp && act == ACT_NONE && (M & 255) == 0 && !colscale && alpha == 1.0f && (ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0))),
                 "gemm(v2): the row-normalising epilogue needs N %% 128 == 0 (N=%d), 16-byte aligned column vectors and either a planes-only output or a plain fp32 output with M %% 256 == 0 (M=%d), no activation / scale", N, M);
  a.rs_out = rs_out; a.rs_strips = N >> 6; a.rn_mr = rn_mr; a.rn_cs = rn_cs;
  a.Ap = Ap; a.lda = lda; a.strideA = strideA;
  a.Wp = Wp; a.strideW = strideW;
  const bool h8c = fmt == MMSA_FMT_H8C;
  a.ldw = h8c ? 3L * K : 2L * K;   // dense packed weights (h8c: row-PAIR stride)
  a.bias = bias; a.strideBias = strideBias; a.colscale = colscale;
  a.resid = resid; a.ldr = ldr; a.strideR = strideR; a.resid_mod = resid_mod; a.beta = beta;
  a.C = C; a.ldc = C ? ldc : 0; a.strideC = strideC;
  a.Cp = Cp; a.ldcp = Cp ? ldcp : 0; a.strideCp = strideCp;
  a.M = M; a.N = N; a.K = K; a.act = act; a.alpha = alpha;
  a.out_mode = out_mode; a.ps_H = ps_H; a.ps_W = ps_W; a.ps_C = ps_C;
  auto log2_exact = [](int v) { int s_ = -1; if (v > 0 && (v & (v - 1)) == 0) { s_ = 0; while ((1 << s_) < v) ++s_; } return s_; };
  a.ps_sw = log2_exact(ps_W); a.ps_sh = log2_exact(ps_H);
  a.cp_fmt = cp_fmt;
  const bool h8 = fmt == MMSA_FMT_H8 || h8c;
  MMSA_CHECK_ARG(!h8 || (K & 63) == 0, "gemm(v2): h8 operands need K %% 64 == 0 (K=%d)", K);
  // workgroup flavour: 8-wave 256-row ping-pong tiles, except shallow contractions on bf16 hi/lo planes (K <= 256: at most four k-tile
  // pairs per output tile -- the tile is its prologue and epilogue), which take the 4-wave flavour (128-row tiles, two workgroups per
  // CU, 2-slot ring: one workgroup's epilogue under the other's k loop; profiles/r03_gemm_flavour4.txt: -9 ... -12 % there, slower on
  // every deeper shape).  `flavour` = 4 / 8 forces one (tests, A/B runs); results are bit-identical either way.  (bf16 planes only.)
  MMSA_CHECK_ARG(flavour == 0 || flavour == 4 || flavour == 8, "gemm(v2): flavour %d (0 = by shape, 4, 8)", flavour);
  // (not under a grid cap: a caller that runs concurrent chains gives each GEMM `max_grid` workgroups so that it holds that many CUs; 2 x
  // max_grid half-size workgroups would be spread over twice as many CUs and their 64 KiB each would shut the other chain's 144 KiB workgroups out)
  // CU count of the current device + the kernels' LDS attributes: set once per DEVICE (common.h mmsa_per_device)
  static MmsaPerDevice per_dev_ = {};
  const int num_cus = mmsa_per_device(per_dev_, [] {
#define V2_ATTR(GEN_, ACT_)                                                                                                   \
  (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<GEN_, ACT_, false, 8, MMSA_FMT_B3>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES(8)); \
  (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_B3>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES(8));  \
  (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_H8>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES(8));  \
  (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_F3>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES(8));  \
  (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<GEN_, ACT_, false, 4, MMSA_FMT_B3>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_BYTES(4));
    V2_ATTR(false, ACT_NONE) V2_ATTR(false, ACT_GELU) V2_ATTR(false, ACT_RELU) V2_ATTR(false, -1) V2_ATTR(true, -1) V2_ATTR(true, ACT_NONE)
#undef V2_ATTR
  });
  // max_grid > 0: at most that many persistent workgroups -- a caller that runs independent chains on concurrent streams gives each
  // GEMM its share of the CUs, so that the kernels of two chains are resident together (one 144 KiB workgroup fits a CU).  The tile
  // shape below is chosen for THAT many CUs (the value of an output element does not depend on the shape of its tile).
  const int cus = (max_grid > 0 && max_grid < num_cus) ? max_grid : num_cus;
  // h8c operands, 4-wave flavour (gemm_h8c4.hip; round 6; OFF by default, see MMSA_H8C4_DEFAULT below): 128 x 128 tiles, two workgroups per CU -- meant for launches whose 256-row tiling is at most ONE round of
  // the CUs it may use (one tile per CU: fill -> k loop -> epilogue with nothing to overlap; proj 8192 x 1024 x 1024 is exactly that) and whose contraction is
  // short enough that the tile is not its k loop (K <= 1024; lin2, K = 4096, measured slower in this form: profiles/r05_h8c_2wg_microbench.txt).  Plain epilogue
  // family only (no activation, no pixel-shuffle / broadcast residual).  `flavour` 4 / 8 force either form (tests, A/B); results are bit-identical.
#ifndef MMSA_H8C4_DEFAULT
#define MMSA_H8C4_DEFAULT 0   // 1 (A/B builds: tools/build_variant.sh ... gemm_v2.hip -DMMSA_H8C4_DEFAULT=1): dispatch the 4-wave flavour by shape.  Measured step-neutral
                              // (profiles/r06_h8c_4wave.txt: proj 58.2 vs 56.8 us, step 32.10 / 32.21 vs 32.29 / 32.17 ms) -- the -18 % of round 5's microbenchmark was its lighter
                              // epilogue; with the site's real one (fp32 rows + planes + strip sums + residual) two resident workgroups pay for it twice on the same issue ports.
                              // So: only when forced (`flavour` = 4: the bit-identity test)
#endif
  bool h8c4 = false;
  if (h8c && out_mode == 0 && resid_mod <= 0 && act == ACT_NONE && flavour != 8 && M >= 128) {
    const long t256 = (long)cdiv(M, 256) * cdiv(N, V2_BN) * batch;
    h8c4 = flavour == 4 || (t256 <= (long)cus && K <= MMSA_KNOB("MMSA_GEMM_H8C4_MAXK", 1024) && MMSA_KNOB("MMSA_GEMM_H8C4", MMSA_H8C4_DEFAULT) != 0);
  }
  const int nw = h8c4 ? 4 : (h8 || rs_out || rn_mr || fmt == MMSA_FMT_F3) ? 8 : flavour ? flavour : (K <= MMSA_KNOB("MMSA_GEMM_NW4_MAXK", 256) && max_grid <= 0 ? 4 : 8);
  const int bm = nw * 32, wg_per_cu = nw == 4 ? 2 : 1;
  a.nbm = cdiv(M, bm);
  a.bn = V2_BN;
  a.nbn = cdiv(N, V2_BN);
  a.ntiles = a.nbm * a.nbn * batch;
#ifdef MMSA_DEBUG_KNOBS
  a.debug = MMSA_KNOB("MMSA_GEMM_DEBUG", 0);
#endif
  // 96-column tiles when they occupy the CUs better: rounds(tiles) x relative tile cost (0.75) against rounds of 128-column tiles.
  // fp32 output only (the planes epilogue writes whole 64-column strips) and no pixel-shuffle / broadcast-residual store.
  if (!Cp && out_mode == 0 && resid_mod <= 0 && N >= 96 && N % 96 == 0 && !rs_out && !h8c) {   // (a ragged last 96-column tile would run the element-wise
    // epilogue: N = 256 -- the ConvFFN fc1 of the extractors -- was routed here and spent 40 of its 92 us in it, profiles/r03_v3_vs_v2.txt)
    const bool no96 = MMSA_KNOB("MMSA_GEMM_NO96", 0) != 0;   // A/B aid (debug-knob builds)
    const int nbn96 = cdiv(N, 96);
    const long t96 = (long)a.nbm * nbn96 * batch;
    const int slots = cus * wg_per_cu;
    const double c128 = (double)cdiv(a.ntiles, slots), c96 = 0.75 * (double)cdiv(t96, slots);
    // same number of column tiles -> nothing to gain from narrower ones (ragged last tile aside)
    if (!no96 && c96 < c128 - 1e-9) {
      a.bn = 96;
      a.nbn = nbn96;
      a.ntiles = (int)t96;
    }
  }
  // blocked tile order (V2_TILE_MN): among the rectangles tm x tn = 32 that tile the (m-tile, n-tile) grid exactly, the one that
  // makes the XCDs fetch the fewest operand bytes -- every activation panel is fetched once per block column, every weight panel
  // once per block row: (nbn / tn) * M * K + (nbm / tm) * N * K; row-major order when none fits.  MMSA_GEMM_ROWMAJOR=1 forces
  // row-major (A/B aid).  Results do not depend on the order.
  const bool rowmajor = MMSA_KNOB("MMSA_GEMM_ROWMAJOR", 0) != 0;
  a.tm = a.tn = 0;
  if (!rowmajor) {
    double best = 0.0;
    for (int tn = 1; tn <= 32; tn <<= 1) {
      const int tm = 32 / tn;
      if (a.nbn % tn != 0 || a.nbm % tm != 0) continue;
      const double cost = (double)(a.nbn / tn) * M + (double)(a.nbm / tm) * N;
      if (a.tm == 0 || cost < best) { best = cost; a.tm = tm; a.tn = tn; }
    }
  }
  // resident workgroups: one per CU (144 KiB LDS) or two (64 KiB each) -- and no more of them than the number of rounds needs: with
  // 168 tiles on 128 CUs two rounds are needed either way, 84 workgroups of 2 tiles finish when 128 workgroups (40 with 2 tiles, 88
  // with 1) do, and leave 44 CUs to whatever runs on the other streams (the concurrent chain, the neck levels)
  const int slots_ = cus * wg_per_cu;
  const int rounds_ = cdiv(a.ntiles, slots_);
  int grid = cdiv(a.ntiles, rounds_);
  // slack stagger (gemm_v2_shared.h): two or more rounds that do not come out even -> one workgroup per slot, the ones with a tile fewer start late.
  // Estimated tile period in shader cycles: the k loop's measured cycles per k-tile (pair) + an epilogue.
  a.stagger = 0;
#ifndef MMSA_GEMM_STAGGER
#define MMSA_GEMM_STAGGER 0   // 1 (A/B builds): on.  Measured no gain (profiles/r05_epilogue_regs.txt): off
#endif
  if (MMSA_GEMM_STAGGER && MMSA_KNOB("MMSA_GEMM_STAGGER", 1) && nw == 8 && rounds_ >= 2 && a.ntiles % slots_ != 0) {
    grid = slots_;
    a.stagger = h8c ? (K >> 6) * 3800 + 16000 : (K >> 5) * (fmt == MMSA_FMT_H8 ? 1700 : 1900) + 16000;
  }
  const bool gen = out_mode != 0 || resid_mod > 0;
  if (h8c4) return mmsa_gemm_h8c4_dispatch(a, grid, stream);
  if (h8c) return mmsa_gemm_h8c_dispatch(a, grid, gen, act, stream);
  const bool pp = MMSA_KNOB("MMSA_GEMM_PP", 1) != 0;   // 0 (debug-knob builds): every wave in phase (A/B timing)
#define V2_LAUNCH(GEN_, ACT_)                                                                                              \
  do {                                                                                                                     \
    if (fmt == MMSA_FMT_H8) hipLaunchKernelGGL((gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_H8>), dim3(grid), dim3(512), V2_LDS_BYTES(8), stream, a);          \
    else if (fmt == MMSA_FMT_F3) hipLaunchKernelGGL((gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_F3>), dim3(grid), dim3(512), V2_LDS_BYTES(8), stream, a);     \
    else if (nw == 4) hipLaunchKernelGGL((gemm_v2_kernel<GEN_, ACT_, false, 4, MMSA_FMT_B3>), dim3(grid), dim3(256), V2_LDS_BYTES(4), stream, a);     \
    else if (pp) hipLaunchKernelGGL((gemm_v2_kernel<GEN_, ACT_, true, 8, MMSA_FMT_B3>), dim3(grid), dim3(512), V2_LDS_BYTES(8), stream, a);     \
    else hipLaunchKernelGGL((gemm_v2_kernel<GEN_, ACT_, false, 8, MMSA_FMT_B3>), dim3(grid), dim3(512), V2_LDS_BYTES(8), stream, a);            \
  } while (0)
  if (gen) {
    if (act == ACT_NONE) V2_LAUNCH(true, ACT_NONE);   // the up-conv / pos-embed GEMMs: unrolled epilogue
    else V2_LAUNCH(true, -1);
  } else {
    switch (act) {
      case ACT_NONE: V2_LAUNCH(false, ACT_NONE); break;
      case ACT_GELU: V2_LAUNCH(false, ACT_GELU); break;
      case ACT_RELU: V2_LAUNCH(false, ACT_RELU); break;
      default: V2_LAUNCH(false, -1); break;
    }
  }
#undef V2_LAUNCH
  MMSA_CHECK_LAUNCH("gemm_split3(v2)");
  return MMSA_OK;
}
