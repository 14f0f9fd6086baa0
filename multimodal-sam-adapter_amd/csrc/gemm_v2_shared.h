// Shared by the LDS-DMA GEMM kernels (gemm_v2.hip: bf16 hi/lo and h8 line planes; gemm_h8c.hip: h8c planes): argument block, tile
// geometry, tile order.  See gemm_v2.hip for the design notes.
#pragma once
#include "common.h"
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) int mx_v8i;       // operand of the block-scaled fp8 MFMA (32 bytes per lane)
typedef __attribute__((ext_vector_type(8))) _Float16 mx_h8;  // operand of the f16 MFMA

struct GemmV2Args {
  const unsigned short* Ap; long lda; long strideA;    // ilv planes (common.h), lda in bf16 units (>= 2K)
  const unsigned short* Wp; long strideW; long ldw;    // ilv planes, row stride ldw >= 2K (bf16 units)
  const float* bias; long strideBias;
  const float* colscale;
  const float* resid; long ldr; long strideR; int resid_mod; float beta;
  float* C; long ldc; long strideC;
  unsigned short* Cp; long ldcp; long strideCp;
  int M, N, K;
  int act; float alpha;
  int out_mode; int ps_H, ps_W, ps_C;
  int ps_sw, ps_sh;   // log2 of ps_W / ps_H when they are powers of two (the pixel-shuffle row mapping then needs no integer division), else -1
  int nbm, nbn, ntiles;
  int tm, tn;  // tile order inside a batch: tm > 0 -> blocks of tm x tn = 32 tiles (see V2_TILE_MN), 0 -> row-major (m-tile, n-tile)
  int bn;      // columns per output tile: 128 (wave tile 64 x 64) or 96 (wave tile 64 x 48: the fourth n-tile of every wave is skipped).  96 when
               // that fills the CUs better: N = 384 gives 3 tiles of 128 (192 tiles on 256 CUs for the ConvNeXt pw2 GEMMs) or 4 of 96 (256 tiles)
  int cp_fmt;  // format of the planes output Cp: MMSA_FMT_B3 (bf16 hi | lo) or MMSA_FMT_H8 (fp16 hi | e5m2 lo, q(hi): common.h), independent of the operands' format
  // LayerNorm folded into a producer / consumer pair of GEMMs (mmsa_gemm_next_extras; IE:396-421: x -> norm -> qkv / lin1):
  //   rs_out: this GEMM (the producer of the residual stream: proj, lin2, the injector's output projection) also writes, per output row
  //           and 64-column strip, the sum and the sum of squares of the fp32 values it stores: rs_out[(row * rs_strips + strip) * 2 + {0,1}]
  //   rn_mr / rn_cs: this GEMM (the consumer: qkv, lin1) runs on the RAW stream's planes against W o w and normalises in its epilogue:
  //           out = rstd_r * (acc - mean_r * cs_n) + bias_n   with (mean_r, rstd_r) = rn_mr[2 r], rn_mr[2 r + 1] and cs_n = rn_cs[n] =
  //           sum_k of the packed weight row n (what every x_k is actually multiplied with)
  float* rs_out; int rs_strips;
  const float* rn_mr; const float* rn_cs;
  float* clamp_max;   // optional clamp watch word (common.h): the planes output's largest |value| beyond the format's range
  int stagger;   // > 0 (set by the launcher, see "slack stagger" there): estimated shader cycles of one tile period; workgroups that walk one tile fewer than the
                 // longest ones start late by a fraction of it (V2_SLACK_STAGGER below)
#ifdef MMSA_DEBUG_KNOBS
  int debug;   // MMSA_GEMM_DEBUG (timing experiments, debug-knob builds only: tools/build_variant.sh -DMMSA_DEBUG_KNOBS): 1 = no global stores, 2 = no epilogue at all, 3 = every k-tile re-reads k-tile 0 (L2-resident operands), 4 = every DMA piece of a wave re-reads the same 1 KiB (L1-resident operand stream), 5 = 4 + 2, 10 = epilogue without its global stores
#endif
};
#ifdef MMSA_DEBUG_KNOBS
#define V2_DBG(a_) ((a_).debug)
#else
#define V2_DBG(a_) 0     // release builds: the timing ablations are compiled out
#endif

#define V2_BN 128
#define V2_BK 32
#define V2_W_BYTES (V2_BN * 128)                        // 16 KiB
// per workgroup flavour (template parameter NW of the kernel): rows per tile 256 / 128, A stage 32 / 16 KiB, ring 3 x 48 / 2 x 32 KiB
#define V2_LDS_BYTES(NW_) (((NW_) == 8 ? 3 : 2) * ((NW_) * 32 * 128 + V2_W_BYTES))
#ifndef V2_FP8_FIRST
#define V2_FP8_FIRST 1   // 0: fp8 and fp16 MFMAs interleaved per output tile (A/B timing)
#endif
#ifndef V2_EXP_NO_FP8
#define V2_EXP_NO_FP8 0   // timing experiment: leave the fp8 cross-term MFMAs out (wrong results)
#endif
#ifndef V2_FAST_STEPS
#define V2_FAST_STEPS 1   // 0: every k-tile runs the general step (A/B timing; the ablation build -DV2_KABL needs it)
#endif
#ifndef V2_SETPRIO
#define V2_SETPRIO 0   // s_setprio(1) around the MFMA chunks: measured no effect on this kernel (same-box A/B)
#endif

// Slack stagger (round 5).  When the tiles do not divide evenly among the persistent workgroups, most workgroups walk one tile fewer than the longest ones and
// have one tile period of slack.  Started together, all workgroups reach their epilogues together, and the epilogues of the adapter-token GEMMs (read-modify-write
// of the 176 MB fp32 token tensor) run at the chip's HBM + Infinity-Cache limit in bursts while the k loops in between use none of it
// (profiles/r05_epilogue_regs.txt: extractor output projection 137 us = 87 us of k loops + 6 x 8.3 us of epilogue bursts at ~8 TB/s).  The workgroups with
// slack therefore start late by (2 i + 1) / 32 of the estimated tile period, i = their index mod 16: their epilogues spread over the others' k loops and nobody
// finishes later than the longest workgroups do anyway.  A pure timing change: which workgroup computes a tile never changes its value.
#define V2_SLACK_STAGGER(a_, rb_, G_)                                                                       \
  if ((a_).stagger > 0) {                                                                                   \
    const int rem_ = (a_).ntiles % (G_);                                                                    \
    if (rem_ > 0 && (rb_) >= rem_) {                                                                        \
      const unsigned long long d_ = ((unsigned long long)(a_).stagger * (unsigned)((((rb_) - rem_) & 15) * 2 + 1)) >> 5;  \
      const unsigned long long t0_ = __builtin_readcyclecounter();                                          \
      while (__builtin_readcyclecounter() - t0_ < d_) __builtin_amdgcn_s_sleep(8);                          \
    }                                                                                                       \
  }

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// GEN = true compiles in the rarely used index arithmetic (pixel-shuffle store, broadcast residual: integer
// divisions per output row); the common epilogue (GEN = false) has none.
// ACT >= 0: the activation is a compile-time constant (the epilogue then has no activation switch and is small enough to be
// unrolled over the four sub-tiles inside the instruction cache); ACT = -1: runtime a.act, rolled epilogue.
// PP = true: "ping-pong" main loop.  The two waves of every SIMD belong to different groups (waves 0-3 / 4-7) that run
// half a k-tile apart: while one group reads its fragments from LDS and issues its share of the LDS-DMA, the other owns
// the MFMA pipe.  With every wave in phase (PP = false) the whole CU first reads LDS (~1000 cycles, MFMA idle) and then
// computes (1536 cycles, LDS idle): MfmaUtil 36-42 %.
#ifdef V2_STAMP   // timing experiment build only (tools/build_variant.sh -DV2_STAMP): cycle stamps of workgroup 0, k-tiles 8..11
__device__ unsigned long long g_v2_stamps[8 * 4 * 10 + 4];   // + {memtime, memrealtime} at start and end of workgroup 0
extern "C" int mmsa_debug_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_v2_stamps), sizeof(unsigned long long) * (8 * 4 * 10 + 4));
}
#define STAMP(i_) if (stamp_on) tt[i_] = __builtin_readcyclecounter();
#define CLK_SAMPLE(o_) if (blockIdx.x == 0 && threadIdx.x == 0) { g_v2_stamps[320 + (o_)] = __builtin_readcyclecounter(); g_v2_stamps[321 + (o_)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define CLK_SAMPLE(o_)
#define STAMP(i_)
#endif

// Tile index inside a batch -> (m-tile, n-tile).  The 32 workgroups that share an XCD (consecutive logical ids) hold 32 consecutive
// tile indices at any time; row-major order made those one row of up to 32 n-tiles, i.e. every XCD streamed the WHOLE weight
// matrix through its L2 per round (counted as fabric traffic: 1.66 x the compulsory bytes over the model's GEMM mix).  Blocked
// order gives an XCD a tm x tn rectangle (8 x 4 for wide N): tm activation panels + tn weight panels per round.
#define V2_TILE_MN(r_, mi_, ni_)                                                 \
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
