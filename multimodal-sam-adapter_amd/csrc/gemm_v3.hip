// split3 GEMM, third kernel for gfx950: the epilogue of output tile t runs INSIDE the k loop of tile t + 1.
//
// Same math, same operand planes and the same LDS image as gemm_v2.hip (see there and gemm_split3.hip for the reference call
// sites: IE:154-167,488,499; TC:107-111; AM:447-451; OPS/modules/ms_deform_attn.py:103-129) -- every output element goes through
// the same MFMA sequence in the same order, so results are bit-identical to gemm_v2's.  What differs is who holds what:
//   * 4 waves per workgroup, ONE per SIMD, 512 registers each: wave tile 128 x 64 (8 x 4 MFMA tiles), the accumulators of the tile
//     being computed (128 registers) AND those of the tile before it (another 128) live in the register file together;
//   * the finished tile's epilogue (bias / activation / scale / residual, the split into operand planes, the LDS transposition,
//     the global stores) is cut into its eight 16-row sub-tiles and sub-tile u is issued after k-step ~ (u + 1) nk / 8 of the
//     NEXT tile: gemm_v2 stops the matrix pipe of every CU for the epilogue and the chip writes 32 MB at once (25-50 % of a
//     GEMM's time); here the stores trickle out beside the operand stream, 4 KiB per k-tile and CU;
//   * no partner wave to hide LDS latency behind, so the fragments of k-tile kt + 1 are read WHILE kt is computed, into the
//     registers the MFMAs have just released (A tile mi right after the MFMAs of row mi; the W tiles and the last A tile are
//     double-buffered by step parity, so that no read is issued after the step's last MFMA);
//   * 3-slot LDS ring as in gemm_v2 (3 x 48 KiB) + 16 KiB of epilogue staging (4 KiB per wave, XOR-swizzled 256-byte rows).
//     Step kt: compute kt from registers, read kt + 1 from slot (kt+1) % 3, LDS-DMA kt + 3 into slot kt % 3 (read during step
//     kt - 1, released by that step's lgkmcnt(0) + barrier); ONE barrier per k-tile, behind a counted vmcnt that leaves the
//     12 DMA pieces of kt + 3 (and the stores issued since) in flight.
// Shapes it takes (the launcher falls back to gemm_v2 otherwise): M % 256 == 0, N % 128 == 0, K % 64 == 0, plain row mapping;
// epilogue kinds: planes-only output with bias and activation (lin1, qkv, the ConvNeXt pw1s), or fp32 output with bias, column
// scale and residual (proj, lin2, pw2, the adapter's projections).
#include "common.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) int v3_v8i;
typedef __attribute__((ext_vector_type(8))) _Float16 v3_h8;
typedef __attribute__((ext_vector_type(4))) int v3_i4;   // a fragment as carried from step to step: as <8 x i16> the backend splits the
                                                           // loop-carried value into halves and rebuilds it with v_perm_b32 before every use

struct GemmV3Args {
  const unsigned short* Ap; long lda; long strideA;
  const unsigned short* Wp; long strideW; long ldw;
  const float* bias; long strideBias;
  const float* colscale;
  const float* resid; long ldr; long strideR; float beta;
  float* C; long ldc; long strideC;
  unsigned short* Cp; long ldcp; long strideCp;
  int M, N, K;
  float alpha;
  int nbm, nbn, ntiles;
  int tm, tn;
  int cp_fmt;
  int debug;   // MMSA_GEMM_DEBUG (timing experiments): 2 = no epilogue at all, 1 = epilogue without its global stores
  int stagger; // delay iterations per wave number at the head of every step (MMSA_V3_STAGGER, default below)
};

#define V3_BM 256
#define V3_BN 128
#define V3_A_BYTES (V3_BM * 128)
#define V3_W_BYTES (V3_BN * 128)
#define V3_STAGE (V3_A_BYTES + V3_W_BYTES)   // 48 KiB
#define V3_STG_OFF (3 * V3_STAGE)
#define V3_LDS_BYTES (V3_STG_OFF + 4 * 4096)  // 160 KiB: all of a CU's LDS
enum { V3_EPI_PLANES = 0, V3_EPI_C = 1 };
#ifndef V3_EXP
#define V3_EXP 0   // compile-time experiments (register-pressure bisection): 1 = no epilogue code
#endif

#define V3_GLDS16(gptr, lptr)                                                                               \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define V3_STR_(x) #x
#define V3_STR(x) V3_STR_(x)
#define V3_WAIT_VM(n_) do { if constexpr ((n_) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else if constexpr ((n_) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); \
    else if constexpr ((n_) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else if constexpr ((n_) == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); \
    else if constexpr ((n_) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); else static_assert((n_) == 0, "vmcnt count"); } while (0)
#define V3_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define V3_NEPI_PLUS_12 (EPI == V3_EPI_C && RESID ? 20 : 16)

// epilogue staging: 16 rows x 256 bytes per wave; the 16-byte chunk c of row r sits at chunk c ^ r (writes of one column group
// by 16 rows and read-backs of whole rows are both conflict-free)
__device__ __forceinline__ int v3_swz(int r, int byte) { return r * 256 + ((((byte >> 4) ^ r) & 15) << 4) + (byte & 15); }

// The epilogue's LDS traffic is written as inline asm.  hipcc orders an ordinary ds_write / ds_read of the staging area behind every
// LDS-DMA still in flight with s_waitcnt vmcnt(0) (it cannot tell the ring slots from the staging rows), which would drain the
// operand stream at every sub-tile; it does not model an asm statement's memory operations, and nothing needs ordering here: the
// staging rows are private to the wave, its LDS instructions execute in order, and the residual DMA is waited for by count.
__device__ __forceinline__ void v3_lds_w64(unsigned addr, uint2 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void v3_lds_w32x2(unsigned addr, unsigned x, unsigned y) {   // x at addr, y at addr + 8
  asm volatile("ds_write2_b32 %0, %1, %2 offset1:2" :: "v"(addr), "v"(x), "v"(y) : "memory");
}
// four 16-byte reads and their wait in ONE statement (early-clobber outputs): the destinations are not visible to the compiler
// before the data has landed
__device__ __forceinline__ void v3_lds_r128x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, uint4& p0, uint4& p1, uint4& p2, uint4& p3) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}

__device__ __forceinline__ void v3_stage_planes4(unsigned stg, int r, int col, const float4 v, int fmt) {
  const int b = (col >> 5) * 128 + (col & 31) * 2;   // byte of the hi values inside the 256-byte row image (2 k-blocks of 128 B)
  if (fmt == MMSA_FMT_H8) {
    uint2 hi; unsigned lo8, qh8;
    h8_split4(v, hi, lo8, qh8);
    v3_lds_w64(stg + v3_swz(r, b), hi);
    const int lb = (col >> 5) * 128 + 64 + (((col & 31) >> 3) << 4) + (col & 7);
    v3_lds_w32x2(stg + v3_swz(r, lb), lo8, qh8);   // lo bytes, and 8 bytes further (same 16-byte chunk) the q(hi) bytes
  } else {
    uint2 hh, ll;
    split4(v, hh, ll);
    v3_lds_w64(stg + v3_swz(r, b), hh);
    v3_lds_w64(stg + v3_swz(r, b + 64), ll);
  }
}

// One LDS-DMA piece: wave-uniform base (SGPR pair) + 32-bit per-lane offset, M0 written next to its use (3 instructions; the builtin
// form costs a 64-bit vector add plus scalar adds for base and M0 -- one wave per SIMD issues ONE instruction per ~4 cycles whatever
// its kind, and the matrix pipe idles while it does).  M0 is not preserved: nothing else in this kernel keeps a value in it.
__device__ __forceinline__ void v3_dma1(unsigned lds, const void* base, unsigned off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1" :: "s"(lds), "s"(base), "v"(off) : "memory");
}

// SCALE: the fp32 epilogue multiplies by a per-column scale and / or alpha (16 more registers)
// RESID: the fp32 epilogue adds beta * residual
template <int FMT, int EPI, int ACT, bool SCALE, bool RESID>
__global__ __launch_bounds__(256, 1) void gemm_v3_kernel(GemmV3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  const int nk = a.K / 32;
  const int np = nk >> 1;   // step pairs per output tile (K % 64 == 0)

  // XCD-aware logical id (as gemm_v2): blocks with equal blockIdx % 8 get consecutive ids, hence neighbouring tiles
  const int G = gridDim.x;
  int rb = blockIdx.x;
  {
    const int xcd = rb & 7, q = G >> 3, r = G & 7;
    rb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (rb >> 3);
  }
  const int my_tiles = (a.ntiles - rb + G - 1) / G;
  if (my_tiles <= 0) return;
  const int total = my_tiles * nk;

#define V3_TILE_MN(r_, mi_, ni_)                                                 \
  do {                                                                           \
    if (a.tm > 0) {                                                              \
      const int blk_ = (r_) >> 5, loc_ = (r_) & 31;                              \
      const int bpr_ = a.nbn / a.tn;                                             \
      const int bi_ = blk_ / bpr_, bj_ = blk_ - bi_ * bpr_;                      \
      const int lm_ = loc_ / a.tn;                                               \
      mi_ = bi_ * a.tm + lm_;                                                    \
      ni_ = bj_ * a.tn + (loc_ - lm_ * a.tn);                                    \
    } else {                                                                     \
      mi_ = (r_) / a.nbn;                                                        \
      ni_ = (r_) - mi_ * a.nbn;                                                  \
    }                                                                            \
  } while (0)

  // ---- LDS-DMA: one instruction = 8 rows x 128 B; lane -> (row = lane >> 3, slot = lane & 7); the 16-byte piece fetched for LDS
  //      slot s of row r is piece s ^ ((r >> 1) & 7) (gemm_v2's image).  A wave stages 64 activation rows (8 instructions) and 32
  //      weight rows (4) per k-tile.
  const int drow = lane >> 3;
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 8;
  const unsigned voA0 = (unsigned)(drow * (int)a.lda + dpiece) * 2u, voA1 = (unsigned)((drow + 8) * (int)a.lda + (dpiece ^ 32)) * 2u;   // pieces 0 / 1
  const unsigned voW0 = (unsigned)(drow * (int)a.ldw + dpiece) * 2u, voW1 = (unsigned)((drow + 8) * (int)a.ldw + (dpiece ^ 32)) * 2u;
  const unsigned smem_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address of the ring
  const unsigned lds_a = smem_a + wave * 64 * 128, lds_w = smem_a + V3_A_BYTES + wave * 32 * 128;
  const unsigned lda8 = (unsigned)(a.lda * 16), ldw8 = (unsigned)(a.ldw * 16);   // bytes per 8 rows
  const unsigned char *pA, *pW;   // k-tile under the prefetch cursor: first activation / weight row of this wave's share
#define V3_SET_SRC(tile_)                                                        \
  do {                                                                           \
    const int t_ = (tile_);                                                      \
    const int per_b_ = a.nbm * a.nbn;                                            \
    const int bz_ = t_ / per_b_;                                                 \
    const int r_ = t_ - bz_ * per_b_;                                            \
    int tmi_, tni_;                                                              \
    V3_TILE_MN(r_, tmi_, tni_);                                                  \
    pA = reinterpret_cast<const unsigned char*>(a.Ap + (long)bz_ * a.strideA + (long)(tmi_ * V3_BM + wave * 64) * a.lda);  \
    pW = reinterpret_cast<const unsigned char*>(a.Wp + (long)bz_ * a.strideW + (long)(tni_ * V3_BN + wave * 32) * a.ldw);  \
  } while (0)

  int pf_tile = rb, pf_kt = 0, pf_j = 0;
  unsigned pf_off = 0;   // byte offset of the cursor's ring slot
  V3_SET_SRC(pf_tile);
  // the 12 pieces of the k-tile under the cursor: activation rows 8 p .. 8 p + 7 of this wave's share (p = 0..7), weight rows 8 q ..
  // (q = 0..3).  Per step: running lane offsets for even / odd pieces (laundered copies, so that the twelve sums are not kept in
  // registers across the loop), one vector add and one scalar add per piece.
#define V3_STEP_PF_BEGIN()                                                                                  \
    unsigned oe_ = voA0, oo_ = voA1, we_ = voW0, wo_ = voW1;                                                \
    asm volatile("" : "+v"(oe_), "+v"(oo_), "+v"(we_), "+v"(wo_));                                          \
    const unsigned la_ = lds_a + pf_off, lw_ = lds_w + pf_off;
#define V3_PIECE_A(p_)                                                                                      \
  {                                                                                                         \
    if ((p_) & 1) { v3_dma1(la_ + (p_) * 1024, pA, oo_); oo_ += 2 * lda8; }                                 \
    else { v3_dma1(la_ + (p_) * 1024, pA, oe_); oe_ += 2 * lda8; }                                          \
  }
#define V3_PIECE_W(q_)                                                                                      \
  {                                                                                                         \
    if ((q_) & 1) { v3_dma1(lw_ + (q_) * 1024, pW, wo_); wo_ += 2 * ldw8; }                                 \
    else { v3_dma1(lw_ + (q_) * 1024, pW, we_); we_ += 2 * ldw8; }                                          \
  }
// The cursor stops on the job's last k-tile: the steps behind it request that k-tile again, into the slot that is free anyway, so
// that EVERY step issues exactly 12 pieces -- no branch around the pieces, one vmcnt count for every step.
#define V3_PF_ADVANCE()                                     \
  do {                                                      \
    pf_off = pf_off == 2 * V3_STAGE ? 0 : pf_off + V3_STAGE; \
    if (pf_j + 1 < total) {                                 \
      ++pf_j;                                               \
      pA += 128;                                            \
      pW += 128;                                            \
      if (++pf_kt == nk) {                                  \
        pf_kt = 0;                                          \
        pf_tile += G;                                       \
        V3_SET_SRC(pf_tile);                                \
      }                                                     \
    }                                                       \
  } while (0)
// Wave w starts every step w * stagger delay iterations late, so that the four waves' pieces reach the CU's vector-memory front end
// one after the other instead of together (it moves ~43 B/clk whoever asks, a wave issues in order, and while a piece waits to be
// accepted the wave issues no MFMA either).
#define V3_STAGGER() { _Pragma("unroll 1") for (int d_ = wave * a.stagger; d_ > 0; --d_) asm volatile("s_nop 7"); }
#define V3_PF_ALL()                                                                             \
  do {                                                                                          \
    V3_STEP_PF_BEGIN()                                                                          \
    V3_PIECE_A(0) V3_PIECE_A(1) V3_PIECE_A(2) V3_PIECE_A(3) V3_PIECE_A(4) V3_PIECE_A(5) V3_PIECE_A(6) V3_PIECE_A(7) \
    V3_PIECE_W(0) V3_PIECE_W(1) V3_PIECE_W(2) V3_PIECE_W(3)                                     \
    V3_PF_ADVANCE();                                                                            \
  } while (0)

  // ---- fragments (gemm_v2's image: row l15 of a 16-row tile, hi chunk g at slot g ^ ((row >> 1) & 7), lo chunk at slot ^ 4)
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag_hi = l15 * 128 + fslot * 16;
  const int frag_lo = l15 * 128 + (fslot ^ 4) * 16;
  const int fa = wm * 128 * 128;
  const int fw = V3_A_BYTES + wn * 64 * 128;
#define LDA_HI(b_, mi_) (*reinterpret_cast<const v3_i4*>((b_) + fa + (mi_) * 2048 + frag_hi))
#define LDA_LO(b_, mi_) (*reinterpret_cast<const v3_i4*>((b_) + fa + (mi_) * 2048 + frag_lo))
#define LDW_HI(b_, ni_) (*reinterpret_cast<const v3_i4*>((b_) + fw + (ni_) * 2048 + frag_hi))
#define LDW_LO(b_, ni_) (*reinterpret_cast<const v3_i4*>((b_) + fw + (ni_) * 2048 + frag_lo))
#define MX_LO(dst_, src_)  { const v3_i4 u_ = src_; dst_[0] = u_[0]; dst_[1] = u_[1]; dst_[2] = u_[2]; dst_[3] = u_[3]; }
#define MX_HI(dst_, src_)  { const v3_i4 u_ = src_; dst_[4] = u_[0]; dst_[5] = u_[1]; dst_[6] = u_[2]; dst_[7] = u_[3]; }

  f32x4 acc[4][8], accB[4][8];   // [ni][mi]: the tile being computed, the tile being written out
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  v3_i4 ah[7], al[7];            // activation tiles 0..6 (hi, and for bf16 planes lo)
  v3_i4 a7h[2], a7l[2];          // activation tile 7, by step parity
  v3_i4 wh[2][4], wl[2][4];      // weight tiles, by step parity
  v3_v8i opA[8] = {}, opW[4] = {};   // h8: fp8 operand tuples of a k-tile pair ([0..3] = lo chunk of the even k-tile, [4..7] = of the odd one)

  // ---- prologue: k-tiles 0, 1, 2 requested; k-tile 0 into the registers; k-tile 1 visible
  V3_PF_ALL();
  V3_PF_ALL();
  V3_PF_ALL();
  V3_WAIT_VM(24);
  __builtin_amdgcn_s_barrier();
  {
    const unsigned char* b0 = smem;
#pragma unroll
    for (int mi = 0; mi < 7; ++mi) {
      ah[mi] = LDA_HI(b0, mi);
      if constexpr (FMT == MMSA_FMT_H8) { MX_LO(opA[mi], LDA_LO(b0, mi)) } else al[mi] = LDA_LO(b0, mi);
    }
    a7h[0] = LDA_HI(b0, 7);
    if constexpr (FMT == MMSA_FMT_H8) { MX_LO(opA[7], LDA_LO(b0, 7)) } else a7l[0] = LDA_LO(b0, 7);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      wh[0][ni] = LDW_HI(b0, ni);
      if constexpr (FMT == MMSA_FMT_H8) { MX_LO(opW[ni], LDW_LO(b0, ni)) } else wl[0][ni] = LDW_LO(b0, ni);
    }
  }
  V3_WAIT_LGKM0();
  V3_WAIT_VM(12);
  __builtin_amdgcn_s_barrier();

  unsigned rd_off = V3_STAGE;   // byte offset of the slot the running step reads (k-tile kt + 1)
  int young = 0;

  // end of a step: my reads of slot (kt+1) % 3 are done, the DMAs of k-tile kt + 2 (issued one step ago) have landed, barrier
// (`young`: an epilogue sub-tile -- its 4 stores and, with a residual, the 4 DMA requests of the next sub-tile's rows -- was issued since
// the last step: they are YOUNGER than the k-tile this wait is for, so the count leaves them in flight too and they get two steps
// to complete instead of one)
#define V3_STEP_END()                                                       \
  {                                                                          \
    V3_WAIT_LGKM0();                                                         \
    if (young) { V3_WAIT_VM(V3_NEPI_PLUS_12); young = 0; } else V3_WAIT_VM(12); \
    __builtin_amdgcn_s_barrier();                                            \
    rd_off = rd_off == 2 * V3_STAGE ? 0 : rd_off + V3_STAGE;                 \
  }
#define BF8(x_) __builtin_bit_cast(bf16x8, x_)
#define MFMA_B3(ni_, mi_, AH_, AL_, P_)                                                                     \
  acc[ni_][mi_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF8(wl[P_][ni_]), BF8(AH_), acc[ni_][mi_], 0, 0, 0);  \
  acc[ni_][mi_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF8(wh[P_][ni_]), BF8(AL_), acc[ni_][mi_], 0, 0, 0);  \
  acc[ni_][mi_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF8(wh[P_][ni_]), BF8(AH_), acc[ni_][mi_], 0, 0, 0);
#define MFMA_F16(ni_, mi_, AH_, P_)                                                                         \
  acc[ni_][mi_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v3_h8, wh[P_][ni_]), __builtin_bit_cast(v3_h8, AH_), acc[ni_][mi_], 0, 0, 0);
#define MFMA_FP8(ni_, mi_)                                                                                  \
  acc[ni_][mi_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(opW[ni_], opA[mi_], acc[ni_][mi_], 1, 1, 0, MMSA_H8_MFMA_SCALE, 0, 0x7f7f7f7f);

  // ---- bf16 hi/lo step, parity P (literal): 96 MFMAs
#define B3_STEP(P, W)                                                                                       \
  {                                                                                                         \
    const unsigned char* nb = smem + rd_off;                                                                \
    V3_STAGGER()                                                                                            \
    V3_STEP_PF_BEGIN()                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 7; ++mi) {                                                      \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_B3(ni, mi, ah[mi], al[mi], P) }               \
      ah[mi] = LDA_HI(nb, mi);                                                                              \
      al[mi] = LDA_LO(nb, mi);                                                                              \
      if (mi == 0) {                                                                                        \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { wh[1 - P][ni] = LDW_HI(nb, ni); wl[1 - P][ni] = LDW_LO(nb, ni); } \
        a7h[1 - P] = LDA_HI(nb, 7);                                                                         \
        a7l[1 - P] = LDA_LO(nb, 7);                                                                         \
      }                                                                                                     \
      V3_PIECE_A(mi) if (mi < 4) V3_PIECE_W(mi)                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_B3(ni, 7, a7h[P], a7l[P], P) }                  \
    V3_PIECE_A(7)                                                                                           \
    V3_PF_ADVANCE();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    V3_STEP_END()                                                                                           \
  }

  // ---- h8 steps.  Even k-tile (S0): 32 fp16 MFMAs; meanwhile the odd k-tile's fragments and its halves of the fp8 tuples are read.
#define H8_STEP0(W)                                                                                         \
  {                                                                                                         \
    const unsigned char* nb = smem + rd_off;                                                                \
    V3_STAGGER()                                                                                            \
    V3_STEP_PF_BEGIN()                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 7; ++mi) {                                                      \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_F16(ni, mi, ah[mi], 0) }                      \
      ah[mi] = LDA_HI(nb, mi);                                                                              \
      MX_HI(opA[mi], LDA_LO(nb, mi))                                                                        \
      if (mi == 0) {                                                                                        \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { wh[1][ni] = LDW_HI(nb, ni); MX_HI(opW[ni], LDW_LO(nb, ni)) } \
        a7h[1] = LDA_HI(nb, 7);                                                                             \
        MX_HI(opA[7], LDA_LO(nb, 7))                                                                        \
      }                                                                                                     \
      V3_PIECE_A(mi) if (mi < 4) V3_PIECE_W(mi)                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_F16(ni, 7, a7h[0], 0) }                         \
    V3_PIECE_A(7)                                                                                           \
    V3_PF_ADVANCE();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    V3_STEP_END()                                                                                           \
  }
  // Odd k-tile (S1): the 32 block-scaled fp8 MFMAs of the pair first (per output tile the order fp8 -> fp16 of gemm_v2), then the 32
  // fp16 MFMAs; the tuples are free once the fp8 block has been issued, so the next pair's even halves are read under the fp16 block.
#define H8_STEP1(W)                                                                                         \
  {                                                                                                         \
    const unsigned char* nb = smem + rd_off;                                                                \
    V3_STAGGER()                                                                                            \
    V3_STEP_PF_BEGIN()                                                                                      \
    _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) {                                                      \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_FP8(ni, mi) }                                 \
      if (mi == 0) {                                                                                        \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) wh[0][ni] = LDW_HI(nb, ni);                        \
        a7h[0] = LDA_HI(nb, 7);                                                                             \
      }                                                                                                     \
      V3_PIECE_A(mi) if (mi < 4) V3_PIECE_W(mi)                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    _Pragma("unroll") for (int mi = 0; mi < 7; ++mi) {                                                      \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_F16(ni, mi, ah[mi], 1) }                      \
      ah[mi] = LDA_HI(nb, mi);                                                                              \
      MX_LO(opA[mi], LDA_LO(nb, mi))                                                                        \
      if (mi == 0) {                                                                                        \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MX_LO(opW[ni], LDW_LO(nb, ni)) }                 \
        MX_LO(opA[7], LDA_LO(nb, 7))                                                                        \
      }                                                                                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                    \
    }                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) { MFMA_F16(ni, 7, a7h[1], 1) }                         \
    V3_PF_ADVANCE();                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    V3_STEP_END()                                                                                           \
  }
#define V3_PAIR_W(W)                                                         \
  {                                                                          \
    if constexpr (FMT == MMSA_FMT_H8) { H8_STEP0(W) H8_STEP1(W) }            \
    else { B3_STEP(0, W) B3_STEP(1, W) }                                     \
  }
#define V3_PAIR() V3_PAIR_W(0)

  // ---- deferred tile: descriptor + epilogue of one 16-row sub-tile (u literal)
  unsigned char* stg = smem + V3_STG_OFF + wave * 4096;
  const unsigned stg_a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)stg;   // the same as an LDS byte address (asm helpers)
  unsigned short* dCp = nullptr;          // planes output: row (m0 + wm * 128), column strip start, of the deferred tile
  float* dC = nullptr;                    // fp32 output / residual: row (m0 + wm * 128), first column of the wave's strip (wave-uniform)
  const float* dR = nullptr;
  f32x4 bn[4], cn[4];                     // bias (and scale) of the lane's columns 16 ni + 4 g .. + 3
  int d_nb = 0;                           // first column of the wave's strip in the deferred tile
  int since = 0;                          // step pairs since the last residual request (>= 1: 24 younger LDS-DMA pieces, so vmcnt(12) covers it)
#pragma unroll
  for (int i = 0; i < 4; ++i) { bn[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; cn[i] = (f32x4){1.f, 1.f, 1.f, 1.f}; }
  int d_fmt0 = MMSA_CP_BASE(a.cp_fmt), d_fmt1 = d_fmt0;   // planes format of the strip's two 32-column blocks (wave-uniform: the split is a multiple of 32)
  // fp32 epilogue: the residual rows of sub-tile u come in by LDS-DMA, each lane fetching exactly the 16 bytes it will add (row l15,
  // columns 16 ni + 4 g .. + 3) into its own 16 bytes of the staging area: no register destination (nothing the compiler could
  // copy before the data lands), no transposition, no barrier (a lane reads what its own wave requested, behind the wave's vmcnt)
#define V3_RES_DMA(u, l15_, g_)                                                                             \
  {                                                                                                         \
    const float* r_ = dR + ((long)((u) * 16 + (l15_)) * a.ldr + 4 * (g_));                                  \
    _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)                                                        \
      V3_GLDS16(r_ + ni * 16, stg + ni * 1024);                                                             \
    since = 0;                                                                                              \
  }

// (the lane id is laundered through an empty asm at the head of every sub-tile: otherwise the compiler hoists the ~40 staging / store
// addresses of all eight sub-tiles out of the tile loop, holds them across the k loop and spills)
#define V3_EPI(u)                                                                                           \
  {                                                                                                         \
    int lz = lane;                                                                                          \
    asm volatile("" : "+v"(lz));                                                                            \
    const int zl15 = lz & 15, zg = lz >> 4;                                                                 \
    if constexpr (EPI == V3_EPI_PLANES) {                                                                   \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                                    \
        f32x4 t_ = accB[ni][u];                                                                             \
        asm volatile("" : "+v"(t_));                                                                        \
        float4 o = make_float4(t_[0] + bn[ni][0], t_[1] + bn[ni][1], t_[2] + bn[ni][2], t_[3] + bn[ni][3]);  \
        if (ACT == ACT_GELU) o = gelu4(o);                                                                  \
        v3_stage_planes4(stg_a, zl15, ni * 16 + 4 * zg, o, ni < 2 ? d_fmt0 : d_fmt1);                       \
      }                                                                                                     \
      uint4 pk[4];                                                                                          \
      v3_lds_r128x4(stg_a + v3_swz(zg, zl15 * 16), stg_a + v3_swz(zg + 4, zl15 * 16), stg_a + v3_swz(zg + 8, zl15 * 16),  \
                    stg_a + v3_swz(zg + 12, zl15 * 16), pk[0], pk[1], pk[2], pk[3]);                         \
      if (a.debug != 1) {                                                                                   \
        unsigned short* cp_ = dCp + (long)((u) * 16 + zg) * a.ldcp + 8 * zl15;                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
          *reinterpret_cast<uint4*>(cp_ + (long)(4 * i) * a.ldcp) = pk[i];                                  \
      } else { asm volatile("" :: "v"(pk[0].x), "v"(pk[1].x), "v"(pk[2].x), "v"(pk[3].x)); }               \
    } else {                                                                                                \
      if constexpr (RESID) { if (since > 0) V3_WAIT_VM(12); else V3_WAIT_VM(0); }                           \
      float* c_ = dC + ((long)((u) * 16 + zl15) * a.ldc + 4 * zg);                                          \
      uint4 rr[4];                                                                                          \
      if constexpr (RESID) v3_lds_r128x4(stg_a + lz * 16, stg_a + 1024 + lz * 16, stg_a + 2048 + lz * 16, stg_a + 3072 + lz * 16, rr[0], rr[1], rr[2], rr[3]); \
      _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) {                                                    \
        f32x4 t_ = accB[ni][u];                                                                             \
        asm volatile("" : "+v"(t_));                                                                        \
        float4 o = make_float4(t_[0] + bn[ni][0], t_[1] + bn[ni][1], t_[2] + bn[ni][2], t_[3] + bn[ni][3]);  \
        if constexpr (SCALE) { o.x *= cn[ni][0]; o.y *= cn[ni][1]; o.z *= cn[ni][2]; o.w *= cn[ni][3]; }    \
        if constexpr (RESID) {                                                                              \
          /* gemm_v2's rounding: the scaled value first, then ONE fused multiply-add with the residual */    \
          o.x = __builtin_fmaf(a.beta, __uint_as_float(rr[ni].x), o.x); o.y = __builtin_fmaf(a.beta, __uint_as_float(rr[ni].y), o.y); \
          o.z = __builtin_fmaf(a.beta, __uint_as_float(rr[ni].z), o.z); o.w = __builtin_fmaf(a.beta, __uint_as_float(rr[ni].w), o.w); \
        }                                                                                                   \
        if (a.debug != 1) *reinterpret_cast<float4*>(c_ + ni * 16) = o;                                     \
        else { asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w)); }                                \
      }                                                                                                     \
      if constexpr (RESID) { if ((u) < 7) { V3_RES_DMA((u) + 1, zl15, zg) } }                                 \
    }                                                                                                       \
  }
  // macro step u: the k-steps up to ceil((u + 1) np / 8) of the tile being computed, then sub-tile u of the tile before it.  ONE copy of
  // the step pair (the loop over u is rolled); the eight epilogue bodies differ in the accumulator registers they name and are
  // selected by a wave-uniform switch -- eight copies of the pair would not fit the instruction cache.
#define V3_EPI_CASE(u) case u: { V3_EPI(u) young = 1; } break;
#define V3_KEEP_CASE(u) case u: { _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) asm volatile("" :: "v"(accB[ni][u])); } break;

  int tile = rb;
  for (int it = 0; it <= my_tiles; ++it) {
    const bool has_k = it < my_tiles;
    const bool has_epi = it > 0 && a.debug != 2;
    int p = 0;
#pragma unroll 1
    for (int u = 0; u < 8; ++u) {
      const int e_ = has_k ? ((u + 1) * np + 7) >> 3 : 0;
#pragma unroll 1
      for (; p < e_; ++p) { V3_PAIR() ++since; }
      if (has_epi) {
        if (V3_EXP & 1) { switch (u) { V3_KEEP_CASE(0) V3_KEEP_CASE(1) V3_KEEP_CASE(2) V3_KEEP_CASE(3) V3_KEEP_CASE(4) V3_KEEP_CASE(5) V3_KEEP_CASE(6) V3_KEEP_CASE(7) } }
        else { switch (u) { V3_EPI_CASE(0) V3_EPI_CASE(1) V3_EPI_CASE(2) V3_EPI_CASE(3) V3_EPI_CASE(4) V3_EPI_CASE(5) V3_EPI_CASE(6) V3_EPI_CASE(7) } }
      }
    }
    if (has_k) {
      // ---- rotate: the tile just computed becomes the deferred one.  Its column parameters are requested FIRST and waited for behind
      // the 256 register moves (an ordinary load the compiler counts: left pending, every epilogue case would open with vmcnt(0) and
      // drain the operand stream eight times per tile; here the moves cover the latency and the wait is the only one)
      const int per_b = a.nbm * a.nbn;
      const int bz = tile / per_b;
      const int rt = tile - bz * per_b;
      int tmi, tni;
      V3_TILE_MN(rt, tmi, tni);
      const int m0 = tmi * V3_BM + wm * 128;
      d_nb = tni * V3_BN + wn * 64;
      const float* bias = a.bias ? a.bias + (long)bz * a.strideBias : nullptr;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        bn[ni] = bias ? *reinterpret_cast<const f32x4*>(bias + d_nb + ni * 16 + 4 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI == V3_EPI_C && SCALE) {
        const float* colscale = a.colscale ? a.colscale + (long)bz * a.strideBias : nullptr;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          cn[ni] = colscale ? *reinterpret_cast<const f32x4*>(colscale + d_nb + ni * 16 + 4 * g) : (f32x4){1.f, 1.f, 1.f, 1.f};
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) { accB[ni][mi] = acc[ni][mi]; acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      asm volatile("" : "+v"(bn[0]), "+v"(bn[1]), "+v"(bn[2]), "+v"(bn[3]));   // first use of the column parameters: the compiler's wait lands here
      if constexpr (EPI == V3_EPI_PLANES) {
        dCp = a.Cp + (long)bz * a.strideCp + (long)m0 * a.ldcp + 2 * d_nb;   // ilv(d_nb) = 2 d_nb for a multiple of 64
        d_fmt0 = MMSA_CP_AT(a.cp_fmt, d_nb);
        d_fmt1 = MMSA_CP_AT(a.cp_fmt, d_nb + 32);
      } else {
        if constexpr (SCALE) {
          asm volatile("" : "+v"(cn[0]), "+v"(cn[1]), "+v"(cn[2]), "+v"(cn[3]));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) cn[ni] *= a.alpha;
        }
        dC = a.C + (long)bz * a.strideC + (long)m0 * a.ldc + d_nb;
        if constexpr (RESID) {
          dR = a.resid + (long)bz * a.strideR + (long)m0 * a.ldr + d_nb;
          V3_RES_DMA(0, l15, g)
          young = 1;
        }
      }
      tile += G;
    }
  }
  V3_WAIT_VM(0);   // the re-requested last k-tile (V3_PF_ADVANCE) is still landing in this workgroup's LDS
}

static int g_v3_cus = 0;
int g_mmsa_v3_mode = -1;   // set by mmsa_debug_gemm_flavour (gemm_v2.hip): -1 = by MMSA_GEMM_V3 (default on), 0 = off, 1 = on

// Returns 1 when the launch was taken, 0 when the shape / epilogue is not this kernel's (the caller then runs gemm_v2), < 0 on error.
int mmsa_gemm_v3_launch(const unsigned short* Ap, long lda, long strideA,
                        const unsigned short* Wp, long strideW,
                        const float* bias, long strideBias, const float* colscale,
                        const float* resid, long ldr, long strideR, int resid_mod, float beta,
                        float* C, long ldc, long strideC,
                        unsigned short* Cp, long ldcp, long strideCp,
                        int M, int N, int K, int batch, int act, float alpha,
                        int out_mode, int fmt, int cp_fmt, int max_grid, hipStream_t stream) {
  // OFF unless asked for (MMSA_GEMM_V3=1, or mmsa_debug_gemm_flavour(3)): bit-identical to gemm_v2 on every shape it takes, but as measured
  // on MI355X (profiles/r03_v3_vs_v2.txt) its k loop runs 4-20 % behind gemm_v2's ping-pong loop -- one wave per SIMD issues ONE
  // instruction per ~4 cycles, so every LDS-DMA piece, fragment read and epilogue instruction is issue time the MFMAs do not get --
  // and it wins only where gemm_v2's epilogue is latency-bound (fp32 output of few column tiles, one tile per workgroup).
  static const int v3_env = getenv("MMSA_GEMM_V3") ? atoi(getenv("MMSA_GEMM_V3")) : 0;
  const int v3_on = g_mmsa_v3_mode >= 0 ? g_mmsa_v3_mode : v3_env;
  if (!v3_on) return 0;
  if ((M % V3_BM) != 0 || (N % V3_BN) != 0 || (K % 64) != 0 || K < 128 || out_mode != 0 || resid_mod > 0) return 0;
  const bool planes = Cp && !C && !resid && !colscale && alpha == 1.0f && (act == ACT_NONE || act == ACT_GELU) &&
                      (!bias || (((uintptr_t)bias) & 15) == 0) && (strideBias & 3) == 0;
  const bool cout = C && !Cp && act == ACT_NONE && (ldc & 3) == 0 && (strideC & 3) == 0 && (!resid || ((ldr & 3) == 0 && (strideR & 3) == 0)) &&
                    (!bias || (((uintptr_t)bias) & 15) == 0) && (!colscale || (((uintptr_t)colscale) & 15) == 0) && (strideBias & 3) == 0;
  if (!planes && !cout) return 0;
  if (v3_on == 2 && !planes) return 0;   // A/B aid: planes-only epilogues only
  if (v3_on == 3 && !cout) return 0;     // A/B aid: fp32 epilogues only
  GemmV3Args a;
  a.Ap = Ap; a.lda = lda; a.strideA = strideA;
  a.Wp = Wp; a.strideW = strideW; a.ldw = 2L * K;
  a.bias = bias; a.strideBias = strideBias; a.colscale = colscale;
  a.resid = resid; a.ldr = ldr; a.strideR = strideR; a.beta = beta;
  a.C = C; a.ldc = ldc; a.strideC = strideC;
  a.Cp = Cp; a.ldcp = ldcp; a.strideCp = strideCp;
  a.M = M; a.N = N; a.K = K; a.alpha = alpha;
  a.cp_fmt = cp_fmt;
  a.nbm = M / V3_BM; a.nbn = N / V3_BN; a.ntiles = a.nbm * a.nbn * batch;
  static const int dbg = getenv("MMSA_GEMM_DEBUG") ? atoi(getenv("MMSA_GEMM_DEBUG")) : 0;
  a.debug = (dbg == 1 || dbg == 2) ? dbg : 0;
  static const int stg = getenv("MMSA_V3_STAGGER") ? atoi(getenv("MMSA_V3_STAGGER")) : 0;
  a.stagger = stg;
  if (g_v3_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmsa_set_error("gemm_split3(v3): cannot query the device");
      return MMSA_ERR_LAUNCH;
    }
    g_v3_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (getenv("MMSA_GEMM_MAX_GRID")) g_v3_cus = atoi(getenv("MMSA_GEMM_MAX_GRID"));
#define V3_ATTR(F_, E_, A_, S_, R_) (void)hipFuncSetAttribute((const void*)gemm_v3_kernel<F_, E_, A_, S_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS_BYTES);
#define V3_ATTR_F(F_) V3_ATTR(F_, V3_EPI_PLANES, ACT_NONE, false, false) V3_ATTR(F_, V3_EPI_PLANES, ACT_GELU, false, false) \
    V3_ATTR(F_, V3_EPI_C, ACT_NONE, false, false) V3_ATTR(F_, V3_EPI_C, ACT_NONE, false, true) V3_ATTR(F_, V3_EPI_C, ACT_NONE, true, false) V3_ATTR(F_, V3_EPI_C, ACT_NONE, true, true)
    V3_ATTR_F(MMSA_FMT_B3) V3_ATTR_F(MMSA_FMT_H8)
#undef V3_ATTR_F
#undef V3_ATTR
  }
  const int cus = (max_grid > 0 && max_grid < g_v3_cus) ? max_grid : g_v3_cus;
  a.tm = a.tn = 0;
  {
    double best = 0.0;
    for (int tn = 1; tn <= 32; tn <<= 1) {
      const int tm = 32 / tn;
      if (a.nbn % tn != 0 || a.nbm % tm != 0) continue;
      const double cost = (double)(a.nbn / tn) * M + (double)(a.nbm / tm) * N;
      if (a.tm == 0 || cost < best) { best = cost; a.tm = tm; a.tn = tn; }
    }
  }
  const int rounds = cdiv(a.ntiles, cus);
  const int grid = cdiv(a.ntiles, rounds);
  const bool h8 = fmt == MMSA_FMT_H8;
#define V3_LAUNCH(F_, E_, A_, S_, R_) hipLaunchKernelGGL((gemm_v3_kernel<F_, E_, A_, S_, R_>), dim3(grid), dim3(256), V3_LDS_BYTES, stream, a)
#define V3_LAUNCH_F(E_, A_, S_, R_) do { if (h8) V3_LAUNCH(MMSA_FMT_H8, E_, A_, S_, R_); else V3_LAUNCH(MMSA_FMT_B3, E_, A_, S_, R_); } while (0)
  const bool scale = colscale || alpha != 1.0f;
  if (planes) {
    if (act == ACT_GELU) V3_LAUNCH_F(V3_EPI_PLANES, ACT_GELU, false, false); else V3_LAUNCH_F(V3_EPI_PLANES, ACT_NONE, false, false);
  } else if (scale) {
    if (resid) V3_LAUNCH_F(V3_EPI_C, ACT_NONE, true, true); else V3_LAUNCH_F(V3_EPI_C, ACT_NONE, true, false);
  } else {
    if (resid) V3_LAUNCH_F(V3_EPI_C, ACT_NONE, false, true); else V3_LAUNCH_F(V3_EPI_C, ACT_NONE, false, false);
  }
#undef V3_LAUNCH_F
#undef V3_LAUNCH
  MMSA_CHECK_LAUNCH("gemm_split3(v3)");
  return 1;
}
