// h8 GEMM on "h8c" operand planes (round 4): same math, tile shape (256 x 128, 8 waves, wave tile 64 x 64), persistent workgroups, tile
// order and epilogue as gemm_v2.hip's h8 flavour -- see there for the reference call sites (IE:488,499,162-167; AM:447-451;
// ops/modules/ms_deform_attn.py:103-129; BK:324) -- with a different operand layout and k loop.
//
// Why.  gemm_v2's h8 loop is bound by the L2 -> LDS operand stream: an LDS-DMA instruction costs per 128-byte line it touches (~30 cycles per
// 1 KiB instruction and CU beside MFMAs: profiles/r03_dma_lanes_microbench.txt, r04_h8c_loop_microbench.txt), a k-tile pair of a 256 x 128
// tile is 96 of them (2880 cycles) against 2048 cycles of MFMA issue, and its matrix phases are uneven (256 / 768 cycles: the 16 fp8 MFMAs
// of a pair all fall into its second k-tile).  Here:
//   * operands are h8c planes (common.h): fp16 hi rows + e5m2 lo bytes packed by row pairs, q(hi) NOT stored (= the top byte of the fp16
//     value, taken in registers with v_perm_b32): 1.5 lines per row and 64-k chunk instead of 2 -> 72 DMA instructions per pair;
//   * a step is a PAIR of k-tiles cut by operand KIND, not by k: phase X reads the hi fragments of both k-tiles (16 ds_read_b128) and
//     issues the 32 fp16 MFMAs, phase Y reads the lo pairs (8 ds_read_b128), builds the fp8 operand tuples and issues the 16 block-scaled
//     fp8 MFMAs: 512 matrix cycles in EVERY phase.  Ping-pong as in gemm_v2: waves 0-3 / 4-7 run one phase apart, one raw s_barrier per phase;
//   * LDS: two HI units (384 rows x 128 B = 48 KiB) + two LO units (384 rows x 64 B = 24 KiB); HI(j+2) is requested in phase Y of pair j
//     (its unit was last read in phase X of pair j), LO(j+1) in phase X of pair j: every piece has >= 4 phases to land.  16 KiB between
//     the LO units make (free LO unit + gap) a 40 KiB epilogue staging area (the epilogue include needs 34 KiB);
//   * scheduling is pinned (the first version lost 25 % to the compiler: tools/exp/h8c_loop.hip header): straight-line steady-state pairs,
//     sched_barrier on both sides of every s_barrier, v_perm_b32 as volatile asm (it was sunk into the MFMA phase); the epilogue sees the
//     lane id through an opaque copy (its address arithmetic, hoisted above the tile loop, was ~200 spilled registers).
// Loop-only timing against gemm_v2's h8 loop (same shapes, profiles/r04_h8c_loop_microbench.txt): K = 4096 -15 %, K = 1024 -8 ... -14 %,
// K = 512 +-3 %, K = 256 +9 % (few pairs per tile: the general first / last pairs dominate) -- the host side packs K >= 512 sites in this format.
#include "gemm_v2_shared.h"

typedef __attribute__((ext_vector_type(4))) unsigned hc_u4;

#define HC_H_UNIT 49152
#define HC_L_UNIT 24576
#define HC_LDS_H(i_) ((i_) * HC_H_UNIT)
#define HC_LDS_L(i_) (2 * HC_H_UNIT + (i_) * (HC_L_UNIT + 16384))
#define HC_LDS_TOTAL (2 * HC_H_UNIT + 2 * HC_L_UNIT + 16384)

#ifndef HC_KEEP
#define HC_KEEP 0   // 1 (A/B builds): the steady-state pairs keep the previous pair's fp8 tuples live past the next fragment reads, so that those never
                    // land in registers queued MFMAs still read.  Worth 20 % in the loop-only microbenchmark (tools/exp/h8c_loop.hip) -- and costs it all back
                    // here: 64 more live registers in the steady state push ~20 loop invariants into scratch, reloaded around the loop in every output
                    // tile (lin1 loop-only 160 us with, 122 us without: profiles/r04_h8c_gemm.txt).  Off.
#endif
#ifdef HC_EPI_STAMP   // timing experiment build only (tools/build_variant.sh ... -DHC_EPI_STAMP; tools/epi_stamps.py): shader-clock stamps of workgroup 0's SECOND tile boundary,
                      // lane 0 of every wave: 0 = the k loop's last barrier passed, 1 = epilogue vectors requested, 2 = they (and everything older) arrived, 3..6 = sub-tile
                      // 0..3 done (its stores issued), 7 = epilogue left, 8 = the barrier behind it passed
__device__ unsigned long long g_hc_estamps[8 * 16];
extern "C" int mmsa_debug_epi_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_hc_estamps), sizeof(g_hc_estamps)); }
#define EPI_STAMP(i_) { if (estamp_on_) { __builtin_amdgcn_sched_barrier(0); et_[i_] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } }
#define EPI_STAMP_WAIT() { if (estamp_on_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#else
#define EPI_STAMP(i_)
#define EPI_STAMP_WAIT()
#endif
#ifndef HC_SETPRIO
#define HC_SETPRIO 0   // 1 (A/B builds): s_setprio 1 around the matrix parts of a pair, 0 around its read / request parts
#endif
#define HC_PRIO(p_) { if (HC_SETPRIO) __builtin_amdgcn_s_setprio(p_); }
#ifndef HC_LATE_DRAIN
#define HC_LATE_DRAIN 0   // 1 (A/B builds): the last pair of an output tile does NOT drain the operand stream -- the next tile's first pieces (requested during that pair) stay in
                          // flight into the epilogue, which waits for them together with its own bias / column / row vectors (gemm_v2_epilogue.inc EPI_DRAIN): one exposed round
                          // trip per tile boundary instead of two.  Measured a wash (lin1 / qkv -1 %, lin2 / proj +1 ... +2 %, step +0.1 ms: profiles/r05_epilogue_regs.txt, job r05_x)
#endif
#ifndef HC_EPI_UNROLL
#define HC_EPI_UNROLL 1   // 0 (A/B builds): the rolled epilogue for every instantiation
#endif
template <bool GEN, int ACT>
__global__ __launch_bounds__(512, 1) void gemm_h8c_kernel(GemmV2Args a) {
  constexpr bool PP = true;
  constexpr bool EPI_UNROLL = ACT >= 0 && HC_EPI_UNROLL;
  constexpr int V2_BM = 256;
  constexpr int V2_NST = 3;   // (epilogue include: unused on the PP path)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, grp = wave >> 2;
  const int l15 = lane & 15, g = lane >> 4;
  const int K = a.K;
  const int np = K >> 6;                      // k-tile pairs per output tile
  const bool ni4 = true;                      // 128-column tiles only
  const int swid = 64;
  const int G = gridDim.x;
  int rb = blockIdx.x;
  { const int xcd = rb & 7, q = G >> 3, r = G & 7; rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3); }
  const int my_tiles = (a.ntiles - rb + G - 1) / G;
  if (my_tiles <= 0) return;
  V2_SLACK_STAGGER(a, rb, G)
  const int total = my_tiles * np;            // pairs this workgroup walks: one continuous DMA stream across its output tiles

  // ---- DMA lane mapping.  HI: one instruction = 8 rows x 128 B (a row's 64-k chunk), lane -> (row = lane >> 3, LDS slot = lane & 7), the
  // piece fetched into slot s of row r is s ^ ((r >> 1) & 7) (the fragment reads' swizzle, as gemm_v2).  LO: one instruction = one 16-row
  // tile = 8 row-pair lines, lane -> (pair jj = lane >> 3, slot = lane & 7), piece = slot ^ f(jj >> 1), f = {0, 3, 2, 1}: with a pair's two
  // rows in one 128-byte LDS row that permutation makes the 16 lanes of every ds_read_b128 lane group hit 16 distinct 16-byte bank groups.
  // A lane's source offset inside an INTERIOR tile does not depend on the tile: six lane constants (below) + scalar piece offsets + the
  // cursor's scalar tile base.  Tiles that overhang M or N clamp their rows per lane, on the fly (general pairs only).  No per-tile lane
  // state: a spilled register reloaded in front of a DMA instruction would put an s_waitcnt vmcnt(0) -- the whole prefetch stream -- there.
  const int drow = lane >> 3;
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 16;
  const int lq = ((lane & 7) ^ ((-(drow >> 1)) & 3)) * 16;
  const int lds_ha = wave * 32 * 128, lds_hw = 32768 + wave * 16 * 128;
  const int lds_la = wave * 2048, lds_lw = 16384 + wave * 1024;
  const unsigned ldaB = (unsigned)(a.lda * 2), ldwB = (unsigned)(a.ldw * 2);   // row-PAIR strides in bytes
  const unsigned K2 = (unsigned)K * 2u, K4 = (unsigned)K * 4u;
  const unsigned LA_E = (unsigned)(drow >> 1) * ldaB + (unsigned)(drow & 1) * K2 + dpiece, LA_O = LA_E ^ 64u;   // (dpiece ^ 64: the sum's bit 6 is dpiece's: every other term is a multiple of 128)
  const unsigned LW_E = (unsigned)(drow >> 1) * ldwB + (unsigned)(drow & 1) * K2 + dpiece, LW_O = LW_E ^ 64u;
  const unsigned LLA = (unsigned)drow * ldaB + K4 + lq, LLW = (unsigned)drow * ldwB + K4 + lq;

  // two prefetch cursors (scalar state only): HI two pairs ahead of the compute cursor, LO one
  const unsigned char *hA, *hW, *lA, *lW;    // first row pair of the cursor's output tile
  int h_m0 = 0, h_n0 = 0, l_m0 = 0, l_n0 = 0;
  bool h_edge = false, l_edge = false;       // the cursor's tile overhangs M or N
  int hp_tile = rb, hp_p = 0, hp_j = 0, lp_tile = rb, lp_p = 0, lp_j = 0;
#define HC_TILE(t_, bz_, m0_, n0_)                                                                          \
  { const int per_b_ = a.nbm * a.nbn; bz_ = (t_) / per_b_; const int r_ = (t_) - bz_ * per_b_; int tmi_, tni_;  \
    V2_TILE_MN(r_, tmi_, tni_); m0_ = tmi_ * 256; n0_ = tni_ * 128; }
#define HC_SET_H(t_)                                                                                        \
  { int bz_; HC_TILE(t_, bz_, h_m0, h_n0)                                                                   \
    hA = reinterpret_cast<const unsigned char*>(a.Ap + (long)bz_ * a.strideA + (long)(h_m0 >> 1) * a.lda);  \
    hW = reinterpret_cast<const unsigned char*>(a.Wp + (long)bz_ * a.strideW + (long)(h_n0 >> 1) * a.ldw);  \
    h_edge = h_m0 + 256 > a.M || h_n0 + 128 > a.N; }
#define HC_SET_L(t_)                                                                                        \
  { int bz_; HC_TILE(t_, bz_, l_m0, l_n0)                                                                   \
    lA = reinterpret_cast<const unsigned char*>(a.Ap + (long)bz_ * a.strideA + (long)(l_m0 >> 1) * a.lda);  \
    lW = reinterpret_cast<const unsigned char*>(a.Wp + (long)bz_ * a.strideW + (long)(l_n0 >> 1) * a.ldw);  \
    l_edge = l_m0 + 256 > a.M || l_n0 + 128 > a.N; }
  HC_SET_H(hp_tile) HC_SET_L(lp_tile)
// interior tile: scalar base + scalar piece offset + lane constant
#define HC_H_ISSUE_FAST()                                                                                   \
  { unsigned char* d_ = smem + HC_LDS_H(hp_j & 1);                                                          \
    const unsigned char* sa_ = hA + (long)hp_p * 128 + (unsigned long)((unsigned)(wave * 16) * ldaB);       \
    const unsigned char* sw_ = hW + (long)hp_p * 128 + (unsigned long)((unsigned)(wave * 8) * ldwB);        \
    GLDS16(sa_ + LA_E, d_ + lds_ha); GLDS16(sa_ + 4u * ldaB + LA_O, d_ + lds_ha + 1024);                     \
    GLDS16(sa_ + 8u * ldaB + LA_E, d_ + lds_ha + 2048); GLDS16(sa_ + 12u * ldaB + LA_O, d_ + lds_ha + 3072); \
    GLDS16(sw_ + LW_E, d_ + lds_hw); GLDS16(sw_ + 4u * ldwB + LW_O, d_ + lds_hw + 1024); }
#define HC_L_ISSUE_FAST()                                                                                   \
  { unsigned char* d_ = smem + HC_LDS_L(lp_j & 1);                                                          \
    const unsigned char* sa_ = lA + (long)lp_p * 128 + (unsigned long)((unsigned)(wave * 16) * ldaB);       \
    const unsigned char* sw_ = lW + (long)lp_p * 128 + (unsigned long)((unsigned)(wave * 8) * ldwB);        \
    GLDS16(sa_ + LLA, d_ + lds_la); GLDS16(sa_ + 8u * ldaB + LLA, d_ + lds_la + 1024);                       \
    GLDS16(sw_ + LLW, d_ + lds_lw); }
// any tile: rows clamped into the matrix per lane (the clamped rows' products are never stored)
#define HC_HOFF(row_, m0_, ld_) (((unsigned)((row_) - (m0_)) >> 1) * (ld_) + ((unsigned)((row_) - (m0_)) & 1u) * K2)
#define HC_H_ISSUE()                                                                                        \
  if (!h_edge) HC_H_ISSUE_FAST() else {                                                                     \
    unsigned char* d_ = smem + HC_LDS_H(hp_j & 1); const long ko_ = (long)hp_p * 128;                       \
    const int ab_ = h_m0 + wave * 32 + drow, wb_ = h_n0 + wave * 16 + drow;                                  \
    GLDS16(hA + ko_ + (unsigned long)(HC_HOFF(min(ab_, a.M - 1), h_m0, ldaB) + dpiece), d_ + lds_ha);        \
    GLDS16(hA + ko_ + (unsigned long)(HC_HOFF(min(ab_ + 8, a.M - 1), h_m0, ldaB) + (dpiece ^ 64)), d_ + lds_ha + 1024); \
    GLDS16(hA + ko_ + (unsigned long)(HC_HOFF(min(ab_ + 16, a.M - 1), h_m0, ldaB) + dpiece), d_ + lds_ha + 2048);       \
    GLDS16(hA + ko_ + (unsigned long)(HC_HOFF(min(ab_ + 24, a.M - 1), h_m0, ldaB) + (dpiece ^ 64)), d_ + lds_ha + 3072); \
    GLDS16(hW + ko_ + (unsigned long)(HC_HOFF(min(wb_, a.N - 1), h_n0, ldwB) + dpiece), d_ + lds_hw);        \
    GLDS16(hW + ko_ + (unsigned long)(HC_HOFF(min(wb_ + 8, a.N - 1), h_n0, ldwB) + (dpiece ^ 64)), d_ + lds_hw + 1024); }
#define HC_H_ADVANCE() { ++hp_j; if (++hp_p == np) { hp_p = 0; hp_tile += G; if (hp_j < total) HC_SET_H(hp_tile) } }
#define HC_L_ISSUE()                                                                                        \
  if (!l_edge) HC_L_ISSUE_FAST() else {                                                                     \
    unsigned char* d_ = smem + HC_LDS_L(lp_j & 1); const long ko_ = (long)lp_p * 128;                       \
    const int aj_ = (l_m0 >> 1) + wave * 16 + drow, wj_ = (l_n0 >> 1) + wave * 8 + drow;                     \
    GLDS16(lA + ko_ + (unsigned long)((unsigned)(min(aj_, (a.M - 1) >> 1) - (l_m0 >> 1)) * ldaB + K4 + lq), d_ + lds_la);          \
    GLDS16(lA + ko_ + (unsigned long)((unsigned)(min(aj_ + 8, (a.M - 1) >> 1) - (l_m0 >> 1)) * ldaB + K4 + lq), d_ + lds_la + 1024); \
    GLDS16(lW + ko_ + (unsigned long)((unsigned)(min(wj_, (a.N - 1) >> 1) - (l_n0 >> 1)) * ldwB + K4 + lq), d_ + lds_lw); }
#define HC_L_ADVANCE() { ++lp_j; if (++lp_p == np) { lp_p = 0; lp_tile += G; if (lp_j < total) HC_SET_L(lp_tile) } }

  // ---- fragment offsets (HI image: the GEMM's LDS image of gemm_v2 with the two k-tiles of a chunk where it has hi | lo)
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag0 = l15 * 128 + fslot * 16, frag1 = l15 * 128 + (fslot ^ 4) * 16;
  const int fha = (wm * 64) * 128, fhw = 32768 + (wn * 64) * 128;
  const int lo_off = 128 * (l15 >> 1) + 16 * ((((l15 & 1) << 2) | g) ^ ((-(l15 >> 2)) & 3));
  const int fla = (wm * 4) * 1024 + lo_off, flw = 16384 + (wn * 4) * 1024 + lo_off;

  f32x4 acc[4][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j_ = 0; j_ < 4; ++j_) acc[i][j_] = (f32x4){0.f, 0.f, 0.f, 0.f};
  hc_u4 ah0[4], ah1[4], wh0[4], wh1[4];
  mx_v8i opA[4], opW[4];

  // ---- prologue: HI(0), HI(1), LO(0) of the stream
  HC_H_ISSUE() HC_H_ADVANCE()
  if (total > 1) { HC_H_ISSUE() HC_H_ADVANCE() }
  HC_L_ISSUE() HC_L_ADVANCE()
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int j = 0, tile = rb, nowait = 0;
  const unsigned psel = 0x07050301u;
#define HC_SB() __builtin_amdgcn_sched_barrier(0)
#define HC_BAR() { HC_SB(); __builtin_amdgcn_s_barrier(); HC_SB(); }
#define HC_PERM(d_, hi_, lo_) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(d_) : "v"(hi_), "v"(lo_), "s"(psel))
#define HC_WAIT(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
// One pair.  Barriers a (end of X's read part), b (end of X's matrix part), c, d likewise for Y.  Visibility: LO(j) of every wave before
// the phase after b / a (group 0 / 1 reads it then), HI(j+1) before the phase after d / c.  Younger than the needed pieces are, in both
// cases, 9 DMA instructions of this wave (3 LO + 6 HI), so every wait is vmcnt(9) and every wave executes all four of them (the two a
// group does not need are satisfied already: no branch between the barriers).
// FAST_ (literal 1): steady state -- both cursors stay inside the output tile.  FAST_ = 0: the general pair -- the first of a tile (after an
// epilogue its waits are skipped: everything it and the next phase X need was drained before the epilogue, and a counted wait would also
// wait for the epilogue's stores), the last two (the cursors wrap to the next output tile; the last pair drains at d), the stream's tail.
#define HC_PAIR(FAST_, KEEP_)                                                                                              \
  {                                                                                                                   \
    const unsigned char* hb = smem + HC_LDS_H(j & 1);                                                                 \
    const unsigned char* lb = smem + HC_LDS_L(j & 1);                                                                 \
    const bool last = !(FAST_) && p == np - 1;                                                                        \
    const bool tail = !(FAST_) && j + 2 >= total;                                                                     \
    const bool skipw = !(FAST_) && nowait > 0;                                                                        \
    /* ======== phase X, read part */                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
      ah0[i] = *reinterpret_cast<const hc_u4*>(hb + fha + i * 2048 + frag0);                                          \
      ah1[i] = *reinterpret_cast<const hc_u4*>(hb + fha + i * 2048 + frag1);                                          \
    }                                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
      wh0[i] = *reinterpret_cast<const hc_u4*>(hb + fhw + i * 2048 + frag0);                                          \
      wh1[i] = *reinterpret_cast<const hc_u4*>(hb + fhw + i * 2048 + frag1);                                          \
    }                                                                                                                 \
    /* the previous pair's fp8 tuples stay live up to here: the reads above must not be allocated to registers queued MFMAs still read */ \
    /* (KEEP_ = 0: the first pair of a tile -- an epilogue lies between, and 64 registers kept live across it would be spilled) */ \
    if (KEEP_) { _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(opA[i]), "v"(opW[i])); }          \
    HC_SB();                                                                                                          \
    const bool do_l = (FAST_) || lp_j < total;                                                                        \
    if (FAST_) { HC_L_ISSUE_FAST() } else if (do_l) { HC_L_ISSUE() }                                                   \
    HC_SB();                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    if (FAST_) HC_WAIT(9); else if (tail) HC_WAIT(0); else if (!skipw) HC_WAIT(9);                                    \
    HC_BAR()                                                                                                          \
    /* ======== phase X, matrix part: 32 fp16 MFMAs (two independent passes: no accumulator is touched twice in a row) */ \
    HC_PRIO(1)                                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh0[ni]), __builtin_bit_cast(mx_h8, ah0[mi]), acc[ni][mi], 0, 0, 0); \
    HC_SB();                                                                                                          \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                \
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mx_h8, wh1[ni]), __builtin_bit_cast(mx_h8, ah1[mi]), acc[ni][mi], 0, 0, 0); \
    HC_SB();                                                                                                          \
    HC_PRIO(0)                                                                                                        \
    if (FAST_) HC_WAIT(9); else if (tail) HC_WAIT(0); else if (!skipw) HC_WAIT(9);                                    \
    HC_BAR()                                                                                                          \
    /* ======== phase Y, read part: lo pairs -> fp8 tuples.  A operand: [q(hi) k-tile 0 | q(hi) k-tile 1 | lo 0 | lo 1], W operand: */ \
    /* [lo 0 | lo 1 | q(hi) 0 | q(hi) 1] -- byte p of A meets byte p of W with the roles crossed: both cross terms of two k-tiles */ \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                   \
      const hc_u4 la_ = *reinterpret_cast<const hc_u4*>(lb + fla + i * 1024);                                         \
      const hc_u4 lw_ = *reinterpret_cast<const hc_u4*>(lb + flw + i * 1024);                                         \
      int a0_, a1_, a2_, a3_, w0_, w1_, w2_, w3_;   /* whole-tuple definitions below: an element-wise assignment would keep the old tuple live (across the epilogue) */ \
      HC_PERM(a0_, ah0[i][1], ah0[i][0]); HC_PERM(a1_, ah0[i][3], ah0[i][2]);                                         \
      HC_PERM(a2_, ah1[i][1], ah1[i][0]); HC_PERM(a3_, ah1[i][3], ah1[i][2]);                                         \
      HC_PERM(w0_, wh0[i][1], wh0[i][0]); HC_PERM(w1_, wh0[i][3], wh0[i][2]);                                         \
      HC_PERM(w2_, wh1[i][1], wh1[i][0]); HC_PERM(w3_, wh1[i][3], wh1[i][2]);                                         \
      opA[i] = (mx_v8i){a0_, a1_, a2_, a3_, (int)la_[0], (int)la_[1], (int)la_[2], (int)la_[3]};                       \
      opW[i] = (mx_v8i){(int)lw_[0], (int)lw_[1], (int)lw_[2], (int)lw_[3], w0_, w1_, w2_, w3_};                       \
    }                                                                                                                 \
    HC_SB();                                                                                                          \
    const bool do_h = (FAST_) || hp_j < total;                                                                        \
    if (FAST_) { HC_H_ISSUE_FAST() } else if (do_h) { HC_H_ISSUE() }                                                   \
    HC_SB();                                                                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    if (FAST_) HC_WAIT(9); else if (tail) HC_WAIT(0); else if (!skipw) HC_WAIT(9);                                    \
    HC_BAR()                                                                                                          \
    /* ======== phase Y, matrix part: 16 block-scaled fp8 MFMAs (K = 128) */                                          \
    HC_PRIO(1)                                                                                                        \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                                  \
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                                \
        acc[ni][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni], opA[mi], acc[ni][mi], 1, 1, 0, MMSA_H8_MFMA_SCALE, 0, 0x7f7f7f7f); \
    HC_SB();                                                                                                          \
    HC_PRIO(0)                                                                                                        \
    /* both cursors move HERE, where few registers are live (the fragments are dead, the MFMAs queued): a cursor that wraps to the next */ \
    /* output tile runs the tile-index arithmetic, whose temporaries on top of a read phase's 192 live registers spilled loop invariants */ \
    if (FAST_) { ++lp_j; ++lp_p; ++hp_j; ++hp_p; } else { if (do_l) HC_L_ADVANCE() if (do_h) HC_H_ADVANCE() }          \
    HC_SB();                                                                                                          \
    if (FAST_) HC_WAIT(9); else if (tail || (last && !HC_LATE_DRAIN)) HC_WAIT(0); else if (!skipw) HC_WAIT(9);        \
    HC_BAR()                                                                                                          \
    if (!(FAST_)) nowait = 0;                                                                                         \
    ++j;                                                                                                              \
  }

  for (int tdone = 0; tdone < my_tiles; ++tdone) {
    if (grp) HC_BAR()
    // the straight-line pairs assume that both cursors sit in THIS output tile and that it is an interior one (lane-constant source
    // offsets); a tile that overhangs M or N runs general pairs throughout
    bool interior;
    { int bz_, m0_, n0_; HC_TILE(tile, bz_, m0_, n0_) interior = m0_ + 256 <= a.M && n0_ + 128 <= a.N; }
    int p = 0;
    HC_PAIR(0, 0)
    ++p;
    if (interior) {
#pragma unroll 1
      for (; p < np - 2; ++p) HC_PAIR(1, HC_KEEP)
      if (hp_p == np) { hp_p = 0; hp_tile += G; if (hp_j < total) HC_SET_H(hp_tile) }   // the straight-line pairs left the HI cursor at the end of this tile
    }
#pragma unroll 1
    for (; p < np; ++p) HC_PAIR(0, 0)
    if (!grp) HC_BAR()
    // ---- tile boundary: (HC_LATE_DRAIN = 0: every DMA issued so far has landed, the last pair drained; 1: the next tile's HI(j+1) / LO(j) pieces may still be in
    // flight -- into the OTHER LO unit and the HI units, never into the staging area -- and the epilogue drains them).  The LO unit of the last pair is free until phase X of
    // the next pair requests LO(j+1) into it -- with the gap behind / before it: 40 KiB of staging for the epilogue
    if (V2_DBG(a) == 2) {   // timing ablation (debug-knob builds): no epilogue
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j_ = 0; j_ < 4; ++j_) { asm volatile("" :: "v"(acc[i][j_])); acc[i][j_] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      if (HC_LATE_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      nowait = 1;
    } else {
      // the epilogue sees the lane id through an opaque copy: everything it derives from it (row / column indices, 64-bit addresses) is
      // computed HERE, per tile.  Hoisted above the tile loop -- what LICM does with them otherwise -- those values are live across the k
      // loops, whose steady state needs 192 registers for accumulators, hi fragments and fp8 tuples alone: ~200 spilled registers, reloaded
      // (scratch loads, one exposed round trip per group) in every epilogue.
#ifdef HC_EPI_STAMP
      unsigned long long et_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      const bool estamp_on_ = blockIdx.x == 0 && tdone == 1;
#endif
      EPI_STAMP(0)
      int lane_o_ = lane;
      asm volatile("" : "+v"(lane_o_));
      const int lane = lane_o_, l15 = lane_o_ & 15, g = lane_o_ >> 4;
#define EPI_STAGING_BASE (smem + (((j - 1) & 1) ? 2 * HC_H_UNIT + HC_L_UNIT : 2 * HC_H_UNIT))
#define EPI_LATE_DRAIN HC_LATE_DRAIN
#include "gemm_v2_epilogue.inc"
#undef EPI_LATE_DRAIN
#undef EPI_STAGING_BASE
      EPI_STAMP(7)
#ifdef HC_EPI_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      HC_BAR()
      EPI_STAMP(8)
      if (estamp_on_ && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q_ = 0; q_ < 9; ++q_) g_hc_estamps[wave * 16 + q_] = et_[q_];
      }
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    HC_BAR()   // the staging area is free again
    tile += G;
  }
}

// Launch (called by mmsa_gemm_v2_launch in gemm_v2.hip, which builds the argument block, the tile order and the grid).
int mmsa_gemm_h8c_dispatch(const GemmV2Args& a, int grid, bool gen, int act, hipStream_t stream) {
  static MmsaPerDevice per_dev_ = {};   // the kernels' LDS attribute, once per device (common.h)
  (void)mmsa_per_device(per_dev_, [] {
#define HC_ATTR(GEN_, ACT_) (void)hipFuncSetAttribute((const void*)gemm_h8c_kernel<GEN_, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, HC_LDS_TOTAL);
    HC_ATTR(false, ACT_NONE) HC_ATTR(false, ACT_GELU) HC_ATTR(false, ACT_RELU) HC_ATTR(false, -1) HC_ATTR(true, -1) HC_ATTR(true, ACT_NONE)
#undef HC_ATTR
  });
#define HC_LAUNCH(GEN_, ACT_) hipLaunchKernelGGL((gemm_h8c_kernel<GEN_, ACT_>), dim3(grid), dim3(512), HC_LDS_TOTAL, stream, a)
  if (gen) {
    if (act == ACT_NONE) HC_LAUNCH(true, ACT_NONE);
    else HC_LAUNCH(true, -1);
  } else {
    switch (act) {
      case ACT_NONE: HC_LAUNCH(false, ACT_NONE); break;
      case ACT_GELU: HC_LAUNCH(false, ACT_GELU); break;
      case ACT_RELU: HC_LAUNCH(false, ACT_RELU); break;
      default: HC_LAUNCH(false, -1); break;
    }
  }
#undef HC_LAUNCH
  MMSA_CHECK_LAUNCH("gemm_split3(h8c)");
  return MMSA_OK;
}
