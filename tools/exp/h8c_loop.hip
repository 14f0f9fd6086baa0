// Loop microbenchmark for the "h8c" operand layout (round 4): is a k loop that moves 3 operand lines per
// 128 k-values of a row (instead of 4) AND balances the matrix pipe over its four phases faster than gemm_v2's h8 loop?
//
// Layout under test (one operand matrix [rows, K], K % 64 == 0, rows even):
//   HI plane  fp16 [rows][K] row-major (row stride ldh elements): hi = fp16(x).
//   LO plane  bytes: row pair j = r >> 1, 64-k chunk c: ONE 128-byte line at (j * ldl + c * 128): [row 2j: 64 B][row 2j+1: 64 B];
//             a row's 64 B = 4 groups g of 16 B = [e5m2(lo * 2^11) of k = 8g..8g+7 of the chunk's first k-tile | same of its second].
//   q(hi) is NOT stored: the e5m2 image of an fp16 value is its top byte (same exponent width), taken in registers with v_perm_b32.
// Per chunk of 64 k a row costs 128 B (hi) + 64 B (lo) = 1.5 lines instead of 2.
//
// Loop: ping-pong as gemm_v2 (two wave groups half a step apart, raw s_barrier per phase), but a step is a PAIR of k-tiles cut by
// operand kind, not by k: phase X reads the hi fragments of both k-tiles (16 ds_read_b128) and runs the 32 fp16 MFMAs (512 matrix
// cycles), phase Y reads the lo pairs (8 ds_read_b128), builds the fp8 tuples (32 v_perm_b32) and runs the 16 block-scaled fp8 MFMAs
// (512 cycles): every phase has the same matrix time (gemm_v2: 256 / 768).  LDS: two HI units of 48 KiB + two LO units of 24 KiB
// (+ 16 KiB between the LO units so that either LO unit + it is a 40 KiB epilogue staging area).
// DMA per wave and pair: 6 HI pieces (two pairs ahead, issued in phase Y) + 3 LO pieces (one pair ahead, issued in phase X);
// NX of the HI pieces can be moved to phase X (balance of the DMA stream; costs prefetch distance).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/h8c_loop.hip -o gpurun_out/h8c_loop
// Run:   h8c_loop check | time
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(8))) _Float16 h8v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

#define H_UNIT 49152
#define L_UNIT 24576
#define LDS_H(i_) ((i_) * H_UNIT)
#define LDS_L(i_) (2 * H_UNIT + (i_) * (L_UNIT + 16384))
#define LDS_TOTAL (2 * H_UNIT + 2 * L_UNIT + 16384)
#define H8_SCALE 0x74747474

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

struct Args {
  const unsigned char* Ah; long ldh;    // bytes per row of the A hi plane
  const unsigned char* Al; long ldl;    // bytes per row PAIR of the A lo plane
  const unsigned char* Wh; long ldwh;
  const unsigned char* Wl; long ldwl;
  float* C; long ldc;                   // optional: plain fp32 result (validation)
  int M, N, K, nbm, nbn, ntiles;
};

#ifndef NX
#define NX 0
#endif
#ifndef ABL
#define ABL 0     // timing ablations (results are garbage): 1 = no MFMAs, 2 = no fragment reads, 4 = no LDS-DMA
#endif
#ifdef STAMPS     // shader clock right after each of the four barriers of pairs 4..11 of workgroup 0's first tile, waves 0 and 4
__device__ unsigned long long g_stamps[2 * 8 * 4 + 4];
#define STAMP(h_) if (blockIdx.x == 0 && tdone == 0 && p >= 4 && p < 12 && (wave & 3) == 0 && lane == 0) g_stamps[grp * 32 + (p - 4) * 4 + (h_)] = __builtin_readcyclecounter();
#else
#define STAMP(h_)
#endif
#ifdef FSTAMPS    // fine stamps inside the phases of pair 6 (waves 0 and 4 of workgroup 0): see FST sites
__device__ unsigned long long g_fst[2 * 16];
#define FST(k_) if (fst_on) ft[k_] = __builtin_readcyclecounter();
#define FST_DECL() unsigned long long ft[16]; const bool fst_on = blockIdx.x == 0 && tdone == 0 && p == 6;
#define FST_STORE() if (fst_on && (wave & 3) == 0 && lane == 0) { _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) g_fst[grp * 16 + q_] = ft[q_]; }
#else
#define FST(k_)
#define FST_DECL()
#define FST_STORE()
#endif

__global__ __launch_bounds__(512, 1) void h8c_loop_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int np = a.K >> 6;
  const int G = gridDim.x;
  int rb = blockIdx.x;
  { const int xcd = rb & 7, q = G >> 3, r = G & 7; rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3); }
  const int my_tiles = (a.ntiles - rb + G - 1) / G;
  if (my_tiles <= 0) return;
  const int total = my_tiles * np;     // pairs this workgroup walks

  // ---- DMA lane mapping
  const int drow = lane >> 3;
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 16;        // HI: byte offset of this lane's piece inside its 128-byte row; odd 8-row groups: ^ 64
  const int ljj = lane >> 3;                                 // LO: row pair inside the 16-row tile
  const int lq = ((lane & 7) ^ ((-(ljj >> 1)) & 3)) * 16;    // LO: byte offset inside the pair's 128-byte line
  const int lds_ha = wave * 32 * 128, lds_hw = 32768 + wave * 16 * 128;
  const int lds_la = wave * 2048, lds_lw = 16384 + wave * 1024;

  // two independent prefetch cursors (HI runs two pairs ahead of the compute cursor, LO one)
  const unsigned char *hA, *hW, *lA, *lW;
  unsigned oa0, oa1, oa2, oa3, ow0, ow1, la0, la1, lw0;
  int hp_tile = rb, hp_p = 0, hp_j = 0, lp_tile = rb, lp_p = 0, lp_j = 0;
#define TILE_MN(t_, m0_, n0_) { const int mi_ = (t_) / a.nbn; m0_ = mi_ * 256; n0_ = ((t_) - mi_ * a.nbn) * 128; }
#define SET_H(t_)                                                                                \
  { int m0_, n0_; TILE_MN(t_, m0_, n0_)                                                          \
    const int ab_ = m0_ + wave * 32 + drow, wb_ = n0_ + wave * 16 + drow;                         \
    hA = a.Ah + (long)m0_ * a.ldh; hW = a.Wh + (long)n0_ * a.ldwh;                                \
    oa0 = (unsigned)((min(ab_, a.M - 1) - m0_) * (int)a.ldh + dpiece);                            \
    oa1 = (unsigned)((min(ab_ + 8, a.M - 1) - m0_) * (int)a.ldh + (dpiece ^ 64));                 \
    oa2 = (unsigned)((min(ab_ + 16, a.M - 1) - m0_) * (int)a.ldh + dpiece);                       \
    oa3 = (unsigned)((min(ab_ + 24, a.M - 1) - m0_) * (int)a.ldh + (dpiece ^ 64));                \
    ow0 = (unsigned)((min(wb_, a.N - 1) - n0_) * (int)a.ldwh + dpiece);                           \
    ow1 = (unsigned)((min(wb_ + 8, a.N - 1) - n0_) * (int)a.ldwh + (dpiece ^ 64)); }
#define SET_L(t_)                                                                                \
  { int m0_, n0_; TILE_MN(t_, m0_, n0_)                                                          \
    const int aj_ = (m0_ >> 1) + wave * 16 + ljj, wj_ = (n0_ >> 1) + wave * 8 + ljj;              \
    lA = a.Al + (long)(m0_ >> 1) * a.ldl; lW = a.Wl + (long)(n0_ >> 1) * a.ldwl;                  \
    la0 = (unsigned)((min(aj_, (a.M - 1) >> 1) - (m0_ >> 1)) * (int)a.ldl + lq);                  \
    la1 = (unsigned)((min(aj_ + 8, (a.M - 1) >> 1) - (m0_ >> 1)) * (int)a.ldl + lq);              \
    lw0 = (unsigned)((min(wj_, (a.N - 1) >> 1) - (n0_ >> 1)) * (int)a.ldwl + lq); }
  SET_H(hp_tile) SET_L(lp_tile)
#define H_PIECE(i_)                                                                              \
  if (!(ABL & 4)) { unsigned char* d_ = smem + LDS_H(hp_j & 1); const long ko_ = (long)hp_p * 128;               \
    if ((i_) == 0) GLDS16(hA + ko_ + oa0, d_ + lds_ha);                                          \
    if ((i_) == 1) GLDS16(hA + ko_ + oa1, d_ + lds_ha + 1024);                                   \
    if ((i_) == 2) GLDS16(hA + ko_ + oa2, d_ + lds_ha + 2048);                                   \
    if ((i_) == 3) GLDS16(hA + ko_ + oa3, d_ + lds_ha + 3072);                                   \
    if ((i_) == 4) GLDS16(hW + ko_ + ow0, d_ + lds_hw);                                          \
    if ((i_) == 5) GLDS16(hW + ko_ + ow1, d_ + lds_hw + 1024); }
#define H_ADVANCE() { ++hp_j; if (++hp_p == np) { hp_p = 0; hp_tile += G; if (hp_j < total) SET_H(hp_tile) } }
#define L_ISSUE()                                                                                \
  if (!(ABL & 4)) { unsigned char* d_ = smem + LDS_L(lp_j & 1); const long ko_ = (long)lp_p * 128;               \
    GLDS16(lA + ko_ + la0, d_ + lds_la); GLDS16(lA + ko_ + la1, d_ + lds_la + 1024);              \
    GLDS16(lW + ko_ + lw0, d_ + lds_lw); }
#define L_ADVANCE() { ++lp_j; if (++lp_p == np) { lp_p = 0; lp_tile += G; if (lp_j < total) SET_L(lp_tile) } }

  // ---- fragment offsets
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag0 = l15 * 128 + fslot * 16, frag1 = l15 * 128 + (fslot ^ 4) * 16;     // k-tile 0 / 1 of the pair
  const int fha = (wm * 64) * 128, fhw = 32768 + (wn * 64) * 128;
  const int lo_off = 128 * (l15 >> 1) + 16 * ((((l15 & 1) << 2) | g) ^ ((-(l15 >> 2)) & 3));
  const int fla = (wm * 4) * 1024 + lo_off, flw = 16384 + (wn * 4) * 1024 + lo_off;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u4v ah0[4], ah1[4], wh0[4], wh1[4];
  v8i opA[4], opW[4];

  // ---- prologue: HI(0), HI(1), LO(0) of the stream
  for (int i = 0; i < 6; ++i) H_PIECE(i)
  H_ADVANCE()
  if (total > 1) { for (int i = 0; i < 6; ++i) H_PIECE(i) H_ADVANCE() }
  L_ISSUE() L_ADVANCE()
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // counted waits: the pieces a phase needs were issued at least one phase earlier; everything younger may stay in flight.  The last pair
  // of an output tile and the last two pairs of the stream (where fewer pieces than assumed are behind the needed ones) drain.
#define NWHI_(n_) n_
#if NX == 0
#define NWHI 9
#elif NX == 1
#define NWHI 8
#elif NX == 2
#define NWHI 7
#else
#define NWHI 6
#endif
#define V_STR_(x) #x
#define V_STR(x) V_STR_(x)
#define WAIT_DMA(n_) { if (last || j + 2 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" V_STR(n_) ")" ::: "memory"); }
#define WAIT_DMA_HI() { if (last || j + 2 >= total) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if (NX == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else if (NX == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
  int j = 0, tile = rb;
  const unsigned psel = 0x07050301u;
#define SB() __builtin_amdgcn_sched_barrier(0)
#define BAR() { SB(); __builtin_amdgcn_s_barrier(); SB(); }
#define PERM(d_, hi_, lo_) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d_) : "v"(hi_), "v"(lo_), "s"(psel))
// One pair.  FAST_ (literal 1): steady state -- both cursors stay inside the output tile, nothing drains: straight-line code between the
// barriers, every wave executes every counted wait (the ones a group does not need are already satisfied).  FAST_ = 0: the general step
// (last two pairs of a tile: the cursors wrap to the next tile, the last pair drains).
#define PAIR(FAST_)                                                                                                   \
  {                                                                                                                   \
    const unsigned char* hb = smem + LDS_H(j & 1);                                                                    \
    const unsigned char* lb = smem + LDS_L(j & 1);                                                                    \
    const bool last = !(FAST_) && p == np - 1;                                                                        \
    const bool drain = !(FAST_) && (last || j + 2 >= total);                                                          \
    FST_DECL()                                                                                                        \
    FST(0)                                                                                                            \
    /* ======== phase X, read part */                                                                                 \
    if (!(ABL & 2)) {                                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
        ah0[i] = *reinterpret_cast<const u4v*>(hb + fha + i * 2048 + frag0);                                          \
        ah1[i] = *reinterpret_cast<const u4v*>(hb + fha + i * 2048 + frag1);                                          \
      }                                                                                                               \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
        wh0[i] = *reinterpret_cast<const u4v*>(hb + fhw + i * 2048 + frag0);                                          \
        wh1[i] = *reinterpret_cast<const u4v*>(hb + fhw + i * 2048 + frag1);                                          \
      }                                                                                                               \
    } else {                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(ah0[i]), "=v"(ah1[i]), "=v"(wh0[i]), "=v"(wh1[i])); \
    }                                                                                                                 \
    /* the fp8 tuples of the previous pair stay live up to here: the fragment reads above must not land in registers that MFMAs still in the queue read */ \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(opA[i]), "v"(opW[i]));                         \
    SB();                                                                                                             \
    FST(1)                                                                                                            \
    if (NX > 0 && ((FAST_) || (hp_j < total && j > 0))) {                                                             \
      _Pragma("unroll") for (int i = 6 - NX; i < 6; ++i) H_PIECE(i)                                                   \
      if (FAST_) { ++hp_j; ++hp_p; } else H_ADVANCE()                                                                 \
    }                                                                                                                 \
    if ((FAST_) || lp_j < total) { L_ISSUE() if (FAST_) { ++lp_j; ++lp_p; } else L_ADVANCE() }                         \
    SB();                                                                                                             \
    FST(2)                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    FST(3)                                                                                                            \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  \
    FST(4)                                                                                                            \
    BAR()                                                                                                             \
    FST(5)                                                                                                            \
    STAMP(0)                                                                                                          \
    /* ======== phase X, matrix part: 32 fp16 MFMAs */                                                                \
    if (!(ABL & 1)) {                                                                                                 \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                              \
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, wh0[ni]), __builtin_bit_cast(h8v, ah0[mi]), acc[ni][mi], 0, 0, 0); \
      SB();                                                                                                           \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                              \
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8v, wh1[ni]), __builtin_bit_cast(h8v, ah1[mi]), acc[ni][mi], 0, 0, 0); \
    } else {                                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(ah0[i]), "v"(ah1[i]), "v"(wh0[i]), "v"(wh1[i])); \
    }                                                                                                                 \
    SB();                                                                                                             \
    FST(6)                                                                                                            \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");  \
    FST(7)                                                                                                            \
    BAR()                                                                                                             \
    FST(8)                                                                                                            \
    STAMP(1)                                                                                                          \
    /* ======== phase Y, read part: lo pairs -> fp8 tuples; q(hi) = top byte of the fp16 hi values */                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
      u4v la, lw;                                                                                                     \
      if (!(ABL & 2)) { la = *reinterpret_cast<const u4v*>(lb + fla + i * 1024); lw = *reinterpret_cast<const u4v*>(lb + flw + i * 1024); } \
      else asm volatile("" : "=v"(la), "=v"(lw));                                                                     \
      PERM(opA[i][0], ah0[i][1], ah0[i][0]); PERM(opA[i][1], ah0[i][3], ah0[i][2]);                                   \
      PERM(opA[i][2], ah1[i][1], ah1[i][0]); PERM(opA[i][3], ah1[i][3], ah1[i][2]);                                   \
      opA[i][4] = (int)la[0]; opA[i][5] = (int)la[1]; opA[i][6] = (int)la[2]; opA[i][7] = (int)la[3];                 \
      opW[i][0] = (int)lw[0]; opW[i][1] = (int)lw[1]; opW[i][2] = (int)lw[2]; opW[i][3] = (int)lw[3];                 \
      PERM(opW[i][4], wh0[i][1], wh0[i][0]); PERM(opW[i][5], wh0[i][3], wh0[i][2]);                                   \
      PERM(opW[i][6], wh1[i][1], wh1[i][0]); PERM(opW[i][7], wh1[i][3], wh1[i][2]);                                   \
    }                                                                                                                 \
    SB();                                                                                                             \
    FST(9)                                                                                                            \
    if ((FAST_) || hp_j < total) {                                                                                    \
      _Pragma("unroll") for (int i = 0; i < 6 - NX; ++i) H_PIECE(i)                                                   \
      if (NX == 0) { if (FAST_) { ++hp_j; ++hp_p; } else H_ADVANCE() }                                                \
    }                                                                                                                 \
    SB();                                                                                                             \
    FST(10)                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    FST(11)                                                                                                           \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" V_STR(NWHI) ")" ::: "memory"); \
    BAR()                                                                                                             \
    FST(12)                                                                                                           \
    STAMP(2)                                                                                                          \
    /* ======== phase Y, matrix part: 16 block-scaled fp8 MFMAs */                                                    \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                \
        if (!(ABL & 1)) acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, H8_SCALE, 0, 0x7f7f7f7f); \
        else asm volatile("" :: "v"(opW[ni]), "v"(opA[mi]));                                                          \
    SB();                                                                                                             \
    FST(13)                                                                                                           \
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(" V_STR(NWHI) ")" ::: "memory"); \
    FST(14)                                                                                                           \
    BAR()                                                                                                             \
    FST(15)                                                                                                           \
    FST_STORE()                                                                                                       \
    STAMP(3)                                                                                                          \
    ++j;                                                                                                              \
  }
  for (int tdone = 0; tdone < my_tiles; ++tdone) {
    if (grp) BAR()
    int p = 0;
    if (NX > 0 && tdone == 0 && np > 2) { PAIR(0) ++p; }
#pragma unroll 1
    for (; p < np - 2; ++p) PAIR(1)
    if (hp_p == np) { hp_p = 0; hp_tile += G; if (hp_j < total) SET_H(hp_tile) }   // the straight-line pairs left the HI cursor at the end of this tile
#pragma unroll 1
    for (; p < np; ++p) PAIR(0)
    if (!grp) BAR()
    // ---- minimal epilogue (validation only): C[m][n] straight from the accumulator layout (lane: m = l15, n = 4g .. 4g+3)
    {
      int m0, n0;
      TILE_MN(tile, m0, n0)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int m = m0 + wm * 64 + mi * 16 + l15, n = n0 + wn * 64 + ni * 16 + 4 * g;
          if (a.C && m < a.M && n + 3 < a.N) *reinterpret_cast<f32x4*>(a.C + (long)m * a.ldc + n) = acc[ni][mi];
          else asm volatile("" :: "v"(acc[ni][mi]));
          acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    tile += G;
  }
}

// ---------------------------------------------------------------------------------------------------------------- host
static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static unsigned char f2e5m2(float x) {   // round to nearest even via fp16
  unsigned short h = f2h(x);
  unsigned r = (unsigned)h + 0x7Fu + ((h >> 8) & 1u);
  return (unsigned char)(r >> 8);
}
static float e5m2f(unsigned char b) { return h2f((unsigned short)(b << 8)); }

struct Packed { std::vector<unsigned char> hi, lo; long ldh, ldl; };
static Packed pack(const std::vector<float>& x, int rows, int K) {
  Packed p;
  const int rp = (rows + 1) / 2, nc = K / 64;
  p.ldh = (long)K * 2; p.ldl = (long)nc * 128;
  p.hi.assign((size_t)rows * K * 2, 0); p.lo.assign((size_t)rp * nc * 128, 0);
  for (int r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) {
      const float v = x[(size_t)r * K + k];
      const unsigned short h = f2h(v);
      memcpy(&p.hi[((size_t)r * K + k) * 2], &h, 2);
      const float lo = (v - h2f(h)) * 2048.f;
      const int c = k >> 6, kk = k & 63, kt = kk >> 5, gq = (kk & 31) >> 3, e = kk & 7;
      p.lo[(size_t)(r >> 1) * p.ldl + c * 128 + (r & 1) * 64 + gq * 16 + kt * 8 + e] = f2e5m2(lo);
    }
  return p;
}

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  hipFuncSetAttribute((const void*)h8c_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
  struct Shape { int M, N, K; const char* name; };
  std::vector<Shape> shapes;
  if (check) shapes = {{512, 256, 256, "small"}, {700, 384, 192, "ragged"}, {256, 128, 64, "one pair"}, {1024, 512, 1024, "deep"}};
  else shapes = {{8192, 4096, 1024, "lin1"}, {8192, 1024, 4096, "lin2"}, {8192, 3072, 1024, "qkv"}, {8192, 1024, 1024, "proj"},
                 {43008, 1024, 512, "ext out"}, {43008, 1024, 256, "ffn fc2"}, {4096, 4096, 1024, "lin1 1img"}, {4096, 1024, 4096, "lin2 1img"}};
  int bad = 0;
  for (auto& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.f; };
    auto nrm = [&]() { float u1 = rnd() + 1e-7f, u2 = rnd(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); };
    for (auto& v : A) v = nrm();
    for (auto& v : W) v = nrm() / sqrtf((float)K);
    Packed pa = pack(A, M, K), pw = pack(W, N, K);
    unsigned char *dAh, *dAl, *dWh, *dWl; float* dC = nullptr;
    hipMalloc(&dAh, pa.hi.size()); hipMalloc(&dAl, pa.lo.size()); hipMalloc(&dWh, pw.hi.size()); hipMalloc(&dWl, pw.lo.size());
    hipMemcpy(dAh, pa.hi.data(), pa.hi.size(), hipMemcpyHostToDevice); hipMemcpy(dAl, pa.lo.data(), pa.lo.size(), hipMemcpyHostToDevice);
    hipMemcpy(dWh, pw.hi.data(), pw.hi.size(), hipMemcpyHostToDevice); hipMemcpy(dWl, pw.lo.data(), pw.lo.size(), hipMemcpyHostToDevice);
    Args a;
    a.Ah = dAh; a.ldh = pa.ldh; a.Al = dAl; a.ldl = pa.ldl; a.Wh = dWh; a.ldwh = pw.ldh; a.Wl = dWl; a.ldwl = pw.ldl;
    a.M = M; a.N = N; a.K = K; a.nbm = (M + 255) / 256; a.nbn = (N + 127) / 128; a.ntiles = a.nbm * a.nbn; a.C = nullptr; a.ldc = N;
    if (check) {
      hipMalloc(&dC, (size_t)M * N * 4); hipMemset(dC, 0xff, (size_t)M * N * 4);
      a.C = dC;
      for (int grid : {256, 3}) {
        const int gsz = a.ntiles < grid ? a.ntiles : grid;
        hipLaunchKernelGGL(h8c_loop_kernel, dim3(gsz), dim3(512), LDS_TOTAL, 0, a);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        std::vector<float> C((size_t)M * N);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        // references: exact product, and the kernel's arithmetic (hi.hi + trunc(hi).q(lo) + q(lo).trunc(hi))
        double e_exact = 0, e_model = 0, nrm2 = 0;
        for (int m = 0; m < M; m += (M > 600 ? 7 : 1))
          for (int n = 0; n < N; n += (N > 300 ? 5 : 1)) {
            double ex = 0, md = 0;
            for (int k = 0; k < K; ++k) {
              const float av = A[(size_t)m * K + k], wv = W[(size_t)n * K + k];
              const unsigned short ha = f2h(av), hw = f2h(wv);
              const float la = e5m2f(f2e5m2((av - h2f(ha)) * 2048.f)) / 2048.f, lw = e5m2f(f2e5m2((wv - h2f(hw)) * 2048.f)) / 2048.f;
              const float qa = h2f(ha & 0xFF00), qw = h2f(hw & 0xFF00);
              ex += (double)av * wv;
              md += (double)h2f(ha) * h2f(hw) + (double)qa * lw + (double)la * qw;
            }
            const double c = C[(size_t)m * N + n];
            e_exact += (c - ex) * (c - ex); e_model += (c - md) * (c - md); nrm2 += ex * ex;
          }
        const double re = sqrt(e_exact / nrm2), rm = sqrt(e_model / nrm2);
        const bool ok = rm < 2e-6 && re < 1e-4;
        printf("check %-9s M=%d N=%d K=%d grid=%d: rel-L2 vs exact %.3e, vs the kernel's arithmetic %.3e %s\n", s.name, M, N, K, gsz, re, rm, ok ? "OK" : "FAIL");
        bad += !ok;
      }
    } else {
      for (int cap : {256, 128}) {
        const int gsz0 = a.ntiles < cap ? a.ntiles : cap;
        const int rounds = (a.ntiles + gsz0 - 1) / gsz0, gsz = (a.ntiles + rounds - 1) / rounds;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(h8c_loop_kernel, dim3(gsz), dim3(512), LDS_TOTAL, 0, a);
        hipEventRecord(e0, 0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(h8c_loop_kernel, dim3(gsz), dim3(512), LDS_TOTAL, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / reps;
#ifdef FSTAMPS
        if (cap == 256 && &s == &shapes[0]) {
          unsigned long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fst), sizeof(h));
          const char* nm[16] = {"start", "X reads issued", "X dma issued", "X lgkm0", "X vmwait(g1)", "barrier a", "X mfma issued", "X vmwait(g0)", "barrier b", "Y reads+perm", "Y dma issued", "Y lgkm0", "barrier c", "Y mfma issued", "Y vmwait(g0)", "barrier d"};
          for (int gq = 0; gq < 2; ++gq) { printf("fine stamps group %d:", gq); for (int i = 1; i < 16; ++i) printf(" [%s +%llu]", nm[i], h[gq * 16 + i] - h[gq * 16 + i - 1]); printf("\n"); }
        }
#endif
#ifdef STAMPS
        if (cap == 256 && &s == &shapes[0]) {
          unsigned long long h[68]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
          for (int gq = 0; gq < 2; ++gq) { printf("stamps group %d (cycles between barriers):", gq); for (int i = 1; i < 32; ++i) printf(" %llu", h[gq * 32 + i] - h[gq * 32 + i - 1]); printf("\n"); }
        }
#endif
        printf("h8c loop NX=%d ABL=%d %-12s M=%5d N=%5d K=%5d grid=%3d (%d tiles): %8.1f us  %6.1f TFLOP/s algorithmic\n", NX, ABL, s.name, M, N, K, gsz, a.ntiles, us,
               2.0 * M * N * K / us / 1e6);
      }
    }
    hipFree(dAh); hipFree(dAl); hipFree(dWh); hipFree(dWl); if (dC) hipFree(dC);
  }
  if (check) printf(bad ? "CHECK FAILED\n" : "CHECK OK\n");
  return bad ? 1 : 0;
}
