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
//   * XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) walk neighbouring (m-tile, n-tile) pairs so
//     activation rows and weight panels are re-read from that XCD's L2 (speed only, never correctness).
#include "common.h"
#include <stdlib.h>

struct GemmV2Args {
  const unsigned short* Ap; long lda; long strideA;    // ilv planes (common.h), lda in bf16 units (>= 2K)
  const unsigned short* Wp; long strideW;              // ilv planes, row stride 2K
  const float* bias; long strideBias;
  const float* colscale;
  const float* resid; long ldr; long strideR; int resid_mod; float beta;
  float* C; long ldc; long strideC;
  unsigned short* Cp; long ldcp; long strideCp;
  int M, N, K;
  int act; float alpha;
  int out_mode; int ps_H, ps_W, ps_C;
  int nbm, nbn, ntiles;
  int debug;   // MMSA_GEMM_DEBUG (timing experiments only): 1 = no global stores, 2 = no epilogue at all, 3 = every k-tile re-reads k-tile 0 (L2-resident operands)
};

#define V2_BM 256
#define V2_BN 128
#define V2_BK 32
#define V2_A_BYTES (V2_BM * 128)                        // 32 KiB: 256 rows x (64 B hi | 64 B lo)
#define V2_W_BYTES (V2_BN * 128)                        // 16 KiB
#define V2_STAGE (V2_A_BYTES + V2_W_BYTES)              // 48 KiB
#define V2_NST 3

#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// GEN = true compiles in the rarely used index arithmetic (pixel-shuffle store, broadcast residual: integer
// divisions per output row); the common epilogue (GEN = false) has none.
template <bool GEN>
__global__ __launch_bounds__(512) void gemm_v2_kernel(GemmV2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
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
  const int total = my_tiles * nk;

  // ---- DMA: one instruction = 8 rows x 128 B; lane -> (row = lane>>3, slot = lane&7); the 16-byte piece fetched
  //      for LDS slot s of row r is piece = s ^ ((r>>1)&7)   (piece 0-3: hi chunks, 4-7: lo chunks)
  const int drow = lane >> 3;
  // rows 8i + drow of a 16-row tile: key = (row_in_16 >> 1) & 7 = (drow >> 1) + 4*(i & 1)
  const int dpiece = ((lane & 7) ^ (drow >> 1)) * 8;         // even 8-row groups; odd groups use dpiece ^ 32
  const int lds_a = wave * 32 * 128;   // this wave's 32 rows of the A image (4 instructions x 8 rows)
  const int lds_w = wave * 16 * 128;   // 16 rows of the W image (2 instructions)
  const unsigned short *sa0, *sa1, *sa2, *sa3, *sw0, *sw1;

#define SET_TILE_SRC(tile_)                                                      \
  do {                                                                           \
    const int t_ = (tile_);                                                      \
    const int per_b_ = a.nbm * a.nbn;                                            \
    const int bz_ = t_ / per_b_;                                                 \
    const int r_ = t_ - bz_ * per_b_;                                            \
    const int m0_ = (r_ / a.nbn) * V2_BM, n0_ = (r_ % a.nbn) * V2_BN;            \
    const int ab_ = m0_ + wave * 32 + drow, wb_ = n0_ + wave * 16 + drow;        \
    const unsigned short* ap_ = a.Ap + (long)bz_ * a.strideA;                    \
    const unsigned short* wp_ = a.Wp + (long)bz_ * a.strideW;                    \
    sa0 = ap_ + (long)min(ab_, a.M - 1) * a.lda + dpiece;                        \
    sa1 = ap_ + (long)min(ab_ + 8, a.M - 1) * a.lda + (dpiece ^ 32);             \
    sa2 = ap_ + (long)min(ab_ + 16, a.M - 1) * a.lda + dpiece;                   \
    sa3 = ap_ + (long)min(ab_ + 24, a.M - 1) * a.lda + (dpiece ^ 32);            \
    sw0 = wp_ + (long)min(wb_, a.N - 1) * 2 * K + dpiece;                        \
    sw1 = wp_ + (long)min(wb_ + 8, a.N - 1) * 2 * K + (dpiece ^ 32);             \
  } while (0)

#define ISSUE_DMA(kt_, st_)                                                       \
  do {                                                                            \
    unsigned char* sb_ = smem + (st_) * V2_STAGE;                                 \
    const int ko_ = a.debug == 3 ? 0 : (kt_) * 64;   /* debug 3: timing experiment, re-read k-tile 0 */ \
    GLDS16(sa0 + ko_, sb_ + lds_a);                                               \
    GLDS16(sa1 + ko_, sb_ + lds_a + 1024);                                        \
    GLDS16(sa2 + ko_, sb_ + lds_a + 2048);                                        \
    GLDS16(sa3 + ko_, sb_ + lds_a + 3072);                                        \
    GLDS16(sw0 + ko_, sb_ + V2_A_BYTES + lds_w);                                  \
    GLDS16(sw1 + ko_, sb_ + V2_A_BYTES + lds_w + 1024);                           \
  } while (0)

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = l15 within a 16-row tile, hi chunk g at slot g ^ ((row>>1)&7), lo at slot ^ 4
  const int fslot = g ^ ((l15 >> 1) & 7);
  const int frag_hi = l15 * 128 + fslot * 16;
  const int frag_lo = l15 * 128 + (fslot ^ 4) * 16;
  const int fa = (wm * 64) * 128;                    // activation rows of this wave, + mi*2048
  const int fw = V2_A_BYTES + (wn * 64) * 128;       // weight rows of this wave, + ni*2048

  // ---- DMA cursor (runs 3 iterations ahead of the compute cursor; ring slot of iteration i is i % 3)
  int pf_tile = rb, pf_kt = 0, pf_st = 0, pf_j = 0;
  SET_TILE_SRC(pf_tile);
#define ADVANCE_PF()                                        \
  do {                                                      \
    pf_st = pf_st == 2 ? 0 : pf_st + 1;                     \
    ++pf_j;                                                 \
    if (++pf_kt == nk) {                                    \
      pf_kt = 0;                                            \
      pf_tile += G;                                         \
      if (pf_j < total) SET_TILE_SRC(pf_tile);              \
    }                                                       \
  } while (0)
#pragma unroll 1
  for (int i = 0; i < 3 && i < total; ++i) {
    ISSUE_DMA(pf_kt, pf_st);
    ADVANCE_PF();
  }

  // ---- fragments are DOUBLE-BUFFERED IN REGISTERS: while the 48 MFMAs of iteration j run on set j&1, the 16
  // ds_read_b128 of iteration j+1 (ring slot (j+1)%3, landed before this iteration's barrier) fill the other set,
  // four reads in front of each 12-MFMA chunk.  Reading a k-tile's fragments right after the barrier that publishes
  // it (previous version) parked all 8 waves on the same ~1000-cycle LDS burst (128 ds_read_b128 per k-tile per CU
  // at 8 cycles each) with an idle matrix pipe; now the matrix pipe starts on registers the moment the barrier opens.
  bf16x8 ah[2][4], al[2][4], wh[2][4], wl[2][4];
#define READ_FRAGS(set, base_, i)                                                          \
  ah[set][i] = *reinterpret_cast<const bf16x8*>((base_) + fa + (i) * 2048 + frag_hi);       \
  al[set][i] = *reinterpret_cast<const bf16x8*>((base_) + fa + (i) * 2048 + frag_lo);       \
  wh[set][i] = *reinterpret_cast<const bf16x8*>((base_) + fw + (i) * 2048 + frag_hi);       \
  wl[set][i] = *reinterpret_cast<const bf16x8*>((base_) + fw + (i) * 2048 + frag_lo);
  {
    // iteration 0's fragments: its 6 DMA instructions are the oldest of the (up to) 18 in flight
    if (total >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (total == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    READ_FRAGS(0, smem, 0) READ_FRAGS(0, smem, 1) READ_FRAGS(0, smem, 2) READ_FRAGS(0, smem, 3)
  }

  int st = 0, kt = 0, tile = rb, nowait = 0;
  for (int j0 = 0; j0 < total; j0 += 2) {
#define P 0
#include "gemm_v2_step.inc"
#undef P
#define P 1
#include "gemm_v2_step.inc"
#undef P
  }
}

static int g_num_cus = 0;

// Internal launcher, called by mmsa_gemm_split3 (gemm_split3.hip) after argument validation when A comes as planes.
int mmsa_gemm_v2_launch(const unsigned short* Ap, long lda, long strideA,
                        const unsigned short* Wp, long strideW,
                        const float* bias, long strideBias, const float* colscale,
                        const float* resid, long ldr, long strideR, int resid_mod, float beta,
                        float* C, long ldc, long strideC,
                        unsigned short* Cp, long ldcp, long strideCp,
                        int M, int N, int K, int batch, int act, float alpha,
                        int out_mode, int ps_H, int ps_W, int ps_C, hipStream_t stream) {
  GemmV2Args a;
  a.Ap = Ap; a.lda = lda; a.strideA = strideA;
  a.Wp = Wp; a.strideW = strideW;
  a.bias = bias; a.strideBias = strideBias; a.colscale = colscale;
  a.resid = resid; a.ldr = ldr; a.strideR = strideR; a.resid_mod = resid_mod; a.beta = beta;
  a.C = C; a.ldc = C ? ldc : 0; a.strideC = strideC;
  a.Cp = Cp; a.ldcp = Cp ? ldcp : 0; a.strideCp = strideCp;
  a.M = M; a.N = N; a.K = K; a.act = act; a.alpha = alpha;
  a.out_mode = out_mode; a.ps_H = ps_H; a.ps_W = ps_W; a.ps_C = ps_C;
  a.nbm = cdiv(M, V2_BM);
  a.nbn = cdiv(N, V2_BN);
  a.ntiles = a.nbm * a.nbn * batch;
  static const int dbg = getenv("MMSA_GEMM_DEBUG") ? atoi(getenv("MMSA_GEMM_DEBUG")) : 0;
  a.debug = dbg;
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
      mmsa_set_error("gemm_split3(v2): cannot query the device");
      return MMSA_ERR_LAUNCH;
    }
    g_num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_NST * V2_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_v2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, V2_NST * V2_STAGE);
  }
  const int grid = a.ntiles < g_num_cus ? a.ntiles : g_num_cus;   // one resident workgroup per CU (144 KiB LDS each)
  if (out_mode != 0 || resid_mod > 0)
    hipLaunchKernelGGL(gemm_v2_kernel<true>, dim3(grid), dim3(512), V2_NST * V2_STAGE, stream, a);
  else
    hipLaunchKernelGGL(gemm_v2_kernel<false>, dim3(grid), dim3(512), V2_NST * V2_STAGE, stream, a);
  MMSA_CHECK_LAUNCH("gemm_split3(v2)");
  return MMSA_OK;
}
