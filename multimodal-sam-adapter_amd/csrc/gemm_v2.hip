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
//   * LDS image row-major [row][32 k] (64 B rows) with the 16-byte k-chunks of a row XOR-permuted by
//     T[(row>>2)&3], T = {0,2,3,1}: every ds_read_b128 fragment read is bank-conflict free, and since a DMA
//     instruction writes 1 KiB linearly (16 rows x 64 B) the permutation is applied on the per-lane SOURCE address;
//   * XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) walk neighbouring (m-tile, n-tile) pairs so
//     activation rows and weight panels are re-read from that XCD's L2 (speed only, never correctness).
#include "common.h"
#include <stdlib.h>

struct GemmV2Args {
  const unsigned short* Ahi; const unsigned short* Alo; long lda; long strideA;
  const unsigned short* Whi; const unsigned short* Wlo; long strideW;
  const float* bias; long strideBias;
  const float* colscale;
  const float* resid; long ldr; long strideR; int resid_mod; float beta;
  float* C; long ldc; long strideC;
  unsigned short* Chi; unsigned short* Clo; long ldcp; long strideCp;
  int M, N, K;
  int act; float alpha;
  int out_mode; int ps_H, ps_W, ps_C;
  int nbm, nbn, ntiles;
  int debug;   // MMSA_GEMM_DEBUG (timing experiments only): 1 = no global stores, 2 = no epilogue at all
};

#define V2_BM 256
#define V2_BN 128
#define V2_BK 32
#define V2_A_PLANE (V2_BM * 64)                         // 16 KiB
#define V2_W_PLANE (V2_BN * 64)                         // 8 KiB
#define V2_STAGE (2 * V2_A_PLANE + 2 * V2_W_PLANE)      // 48 KiB
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

  // ---- DMA: one instruction = 16 rows x 64 B; lane -> (row = lane>>2, slot = lane&3), k-chunk = slot ^ T[(row>>2)&3]
  const int drow = lane >> 2;
  const int swz = (0x1320 >> (((drow >> 2) & 3) * 4)) & 3;  // T = {0,2,3,1}
  const int dchunk = ((lane & 3) ^ swz) * 8;
  const int lds_a = wave * 32 * 64;   // this wave's rows in an A plane (+1024 for the second instruction)
  const int lds_w = wave * 16 * 64;
  const unsigned short *sa_h0, *sa_h1, *sa_l0, *sa_l1, *sw_h, *sw_l;

#define SET_TILE_SRC(tile_)                                                      \
  do {                                                                           \
    const int t_ = (tile_);                                                      \
    const int per_b_ = a.nbm * a.nbn;                                            \
    const int bz_ = t_ / per_b_;                                                 \
    const int r_ = t_ - bz_ * per_b_;                                            \
    const int m0_ = (r_ / a.nbn) * V2_BM, n0_ = (r_ % a.nbn) * V2_BN;            \
    int ar0_ = m0_ + wave * 32 + drow, ar1_ = ar0_ + 16;                         \
    ar0_ = ar0_ < a.M ? ar0_ : a.M - 1;                                          \
    ar1_ = ar1_ < a.M ? ar1_ : a.M - 1;                                          \
    int wr_ = n0_ + wave * 16 + drow;                                            \
    wr_ = wr_ < a.N ? wr_ : a.N - 1;                                             \
    const long ao_ = (long)bz_ * a.strideA + dchunk;                             \
    const long wo_ = (long)bz_ * a.strideW + (long)wr_ * K + dchunk;             \
    sa_h0 = a.Ahi + ao_ + (long)ar0_ * a.lda;                                    \
    sa_h1 = a.Ahi + ao_ + (long)ar1_ * a.lda;                                    \
    sa_l0 = a.Alo + ao_ + (long)ar0_ * a.lda;                                    \
    sa_l1 = a.Alo + ao_ + (long)ar1_ * a.lda;                                    \
    sw_h = a.Whi + wo_;                                                          \
    sw_l = a.Wlo + wo_;                                                          \
  } while (0)

#define ISSUE_DMA(kt_, st_)                                                       \
  do {                                                                            \
    unsigned char* sb_ = smem + (st_) * V2_STAGE;                                 \
    const int ko_ = (kt_) * V2_BK;                                                \
    GLDS16(sa_h0 + ko_, sb_ + lds_a);                                             \
    GLDS16(sa_h1 + ko_, sb_ + lds_a + 1024);                                      \
    GLDS16(sa_l0 + ko_, sb_ + V2_A_PLANE + lds_a);                                \
    GLDS16(sa_l1 + ko_, sb_ + V2_A_PLANE + lds_a + 1024);                         \
    GLDS16(sw_h + ko_, sb_ + 2 * V2_A_PLANE + lds_w);                             \
    GLDS16(sw_l + ko_, sb_ + 2 * V2_A_PLANE + V2_W_PLANE + lds_w);                \
  } while (0)

  f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (row = l15 within a 16-row tile, chunk g, permuted)
  const int fswz = (0x1320 >> (((l15 >> 2) & 3) * 4)) & 3;
  const int frag_off = l15 * 64 + ((g ^ fswz) * 16);
  const int frag_a = (wm * 64) * 64 + frag_off;                       // activation rows, + mi*1024
  const int frag_w = 2 * V2_A_PLANE + (wn * 64) * 64 + frag_off;      // weight rows, + ni*1024

  // ---- prefetch cursor (runs 2 iterations ahead of the compute cursor)
  int pf_tile = rb, pf_kt = 0, pf_st = 0, pf_j = 0;
  SET_TILE_SRC(pf_tile);
#define PREFETCH_NEXT()                                     \
  do {                                                      \
    ISSUE_DMA(pf_kt, pf_st);                                \
    pf_st = pf_st == 2 ? 0 : pf_st + 1;                     \
    ++pf_j;                                                 \
    if (++pf_kt == nk) {                                    \
      pf_kt = 0;                                            \
      pf_tile += G;                                         \
      if (pf_j < total) SET_TILE_SRC(pf_tile);              \
    }                                                       \
  } while (0)
  PREFETCH_NEXT();
  if (total > 1) PREFETCH_NEXT();

  int st = 0, kt = 0, tile = rb, nowait = 0;
  for (int j = 0; j < total; ++j) {
    // the DMAs of iteration j must have landed; those of j+1 (6 instructions, issued later) may stay in flight.
    // vmcnt retires in order and counts stores too, so a wait issued after an epilogue would also wait for that
    // epilogue's whole store burst; instead the epilogue first drains the (older) DMAs of j+1 and j+2 and the next
    // two iterations skip the wait, which gives the stores ~3 k-tiles of MFMA work to retire behind.
    if (nowait > 0) --nowait;
    else if (j + 1 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (pf_j < total) PREFETCH_NEXT();   // ring slot (j+2)%3 was last read in iteration j-1: free after the barrier
    const unsigned char* base = smem + st * V2_STAGE;
    bf16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(base + frag_a + i * 1024);
      al[i] = *reinterpret_cast<const bf16x8*>(base + V2_A_PLANE + frag_a + i * 1024);
      wh[i] = *reinterpret_cast<const bf16x8*>(base + frag_w + i * 1024);
      wl[i] = *reinterpret_cast<const bf16x8*>(base + V2_W_PLANE + frag_w + i * 1024);
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ni], ah[mi], acc[ni][mi], 0, 0, 0);
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], al[mi], acc[ni][mi], 0, 0, 0);
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ni], ah[mi], acc[ni][mi], 0, 0, 0);
      }
    const int st_cur = st;
    st = st == 2 ? 0 : st + 1;
    if (++kt < nk) continue;
    kt = 0;

    if (a.debug == 2) { tile += G; continue; }
    // ---- epilogue of `tile`.  MFMA layout: lane holds C[m = ..+l15][n = ..+4g .. +3].  The column-wise part
    // (bias, activation, alpha, gamma) is applied in registers; each wave then transposes 16 x 64 sub-tiles through
    // the ring slot it has just finished computing from (free until the DMA of iteration j+3), so that residual
    // loads and output stores are FULL 256-byte row segments (4 rows per wave-instruction) instead of 64-byte
    // (fp32) / 32-byte (planes) fragments.
    {
      const int per_b = a.nbm * a.nbn;
      const int bz = tile / per_b;
      const int rt = tile - bz * per_b;
      const int m0 = (rt / a.nbn) * V2_BM, n0 = (rt % a.nbn) * V2_BN;
      const float* bias = a.bias ? a.bias + (long)bz * a.strideBias : nullptr;
      const float* resid = a.resid ? a.resid + (long)bz * a.strideR : nullptr;
      float* C = a.C ? a.C + (long)bz * a.strideC : nullptr;
      unsigned short* Chi = a.Chi ? a.Chi + (long)bz * a.strideCp : nullptr;
      unsigned short* Clo = a.Clo ? a.Clo + (long)bz * a.strideCp : nullptr;
      const bool vec_ok = ((a.ldc & 3) == 0) && (!resid || (a.ldr & 3) == 0) && ((a.N & 3) == 0) && ((a.ldcp & 3) == 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs of iterations j+1, j+2 (older than the stores below)
      __builtin_amdgcn_s_barrier();  // every wave has read its fragments of slot st_cur; j+1, j+2 landed for all
      nowait = 2;
      float* stg = reinterpret_cast<float*>(smem + st_cur * V2_STAGE) + wave * (16 * 68);
      const int nb_ = n0 + wn * 64;
      // column parameters of this lane's 4x4 columns (n = nb_ + ni*16 + 4g + r): loaded ONCE per tile, unconditionally
      // (clamped index), so the element loop below has no loads, no waits and no divergent branches.
      float bv[4][4], cv[4][4];
      if (bias) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[ni][r] = bias[min(nb_ + ni * 16 + 4 * g + r, a.N - 1)];
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[ni][r] = 0.f;
      }
      if (a.colscale) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int ci = min(nb_ + ni * 16 + 4 * g + r, a.N - 1);
            if constexpr (GEN) { if (a.out_mode == 1) ci %= a.ps_C; }
            cv[ni][r] = a.colscale[ci] * a.alpha;
          }
      } else {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
          for (int r = 0; r < 4; ++r) cv[ni][r] = a.alpha;
      }
      // row mapping (destination row / column, residual row); identity unless GEN
      auto map_row = [&](int m, int n, long& drow_, int& dcol, long& rrow) {
        drow_ = m; dcol = n; rrow = m;
        if constexpr (GEN) {
          if (a.out_mode == 1) {
            const int ij = n / a.ps_C;
            dcol = n - ij * a.ps_C;
            const int w_ = m % a.ps_W;
            const int t_ = m / a.ps_W;
            const int h_ = t_ % a.ps_H;
            const int b_ = t_ / a.ps_H;
            drow_ = ((long)(b_ * 2 * a.ps_H + 2 * h_ + (ij >> 1))) * (2 * a.ps_W) + 2 * w_ + (ij & 1);
          }
          rrow = a.resid_mod > 0 ? (long)((int)drow_ % a.resid_mod) : drow_;
        }
      };
      const int rl0 = lane >> 4;            // read-back: lane -> (row = rl0 + 4*i, columns cl .. cl+3)
      const int cl = (lane & 15) * 4;
      const int n = nb_ + cl;
      const bool fast = vec_ok && (nb_ + 64 <= a.N);   // wave-uniform: whole 64-column strip inside N, 16-byte aligned
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int mb = m0 + wm * 64 + mi * 16;
        // residual rows first: 4 independent 16-byte loads in flight, no waits inside the element loops
        float4 rr[4];
        if (fast && resid) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            long drow_, rrow; int dcol;
            map_row(min(mb + rl0 + 4 * i, a.M - 1), n, drow_, dcol, rrow);
            rr[i] = *reinterpret_cast<const float4*>(resid + rrow * a.ldr + dcol);
          }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
          acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r] + bv[ni][r], a.act) * cv[ni][r];
          *reinterpret_cast<float4*>(stg + l15 * 68 + ni * 16 + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (a.debug == 1) continue;
        if (fast) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rl = rl0 + 4 * i;
            float4 o = *reinterpret_cast<const float4*>(stg + rl * 68 + cl);
            const int m = mb + rl;
            if (resid) { o.x += a.beta * rr[i].x; o.y += a.beta * rr[i].y; o.z += a.beta * rr[i].z; o.w += a.beta * rr[i].w; }
            if (m < a.M) {
              long drow_, rrow; int dcol;
              map_row(m, n, drow_, dcol, rrow);
              if (C) *reinterpret_cast<float4*>(C + drow_ * a.ldc + dcol) = o;
              if (Chi) {
                uint2 hh, ll;
                split4(o, hh, ll);
                *reinterpret_cast<uint2*>(Chi + drow_ * a.ldcp + dcol) = hh;
                *reinterpret_cast<uint2*>(Clo + drow_ * a.ldcp + dcol) = ll;
              }
            }
          }
        } else {  // ragged right edge or unaligned leading dimensions: element-wise, rare
          for (int i = 0; i < 4; ++i) {
            const int rl = rl0 + 4 * i;
            const float4 o4 = *reinterpret_cast<const float4*>(stg + rl * 68 + cl);
            const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
            const int m = mb + rl;
            if (m >= a.M) continue;
            for (int r = 0; r < 4; ++r) {
              if (n + r >= a.N) continue;
              long drow_, rrow; int dcol;
              map_row(m, n + r, drow_, dcol, rrow);
              float x = ov[r];
              if (resid) x += a.beta * resid[rrow * a.ldr + dcol];
              if (C) C[drow_ * a.ldc + dcol] = x;
              if (Chi) {
                unsigned short hh, ll;
                split_bf16(x, hh, ll);
                Chi[drow_ * a.ldcp + dcol] = hh;
                Clo[drow_ * a.ldcp + dcol] = ll;
              }
            }
          }
        }
      }
    }
    tile += G;
  }
}

static int g_num_cus = 0;

// Internal launcher, called by mmsa_gemm_split3 (gemm_split3.hip) after argument validation when A comes as planes.
int mmsa_gemm_v2_launch(const unsigned short* Ahi, const unsigned short* Alo, long lda, long strideA,
                        const unsigned short* Whi, const unsigned short* Wlo, long strideW,
                        const float* bias, long strideBias, const float* colscale,
                        const float* resid, long ldr, long strideR, int resid_mod, float beta,
                        float* C, long ldc, long strideC,
                        unsigned short* Chi, unsigned short* Clo, long ldcp, long strideCp,
                        int M, int N, int K, int batch, int act, float alpha,
                        int out_mode, int ps_H, int ps_W, int ps_C, hipStream_t stream) {
  GemmV2Args a;
  a.Ahi = Ahi; a.Alo = Alo; a.lda = lda; a.strideA = strideA;
  a.Whi = Whi; a.Wlo = Wlo; a.strideW = strideW;
  a.bias = bias; a.strideBias = strideBias; a.colscale = colscale;
  a.resid = resid; a.ldr = ldr; a.strideR = strideR; a.resid_mod = resid_mod; a.beta = beta;
  a.C = C; a.ldc = C ? ldc : 0; a.strideC = strideC;
  a.Chi = Chi; a.Clo = Clo; a.ldcp = Chi ? ldcp : 0; a.strideCp = strideCp;
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
