// Windowed SAM attention (window_size <= 14, head_dim 64) with the decomposed relative-position bias FUSED, gfx950.
//
// Reference: Block.forward IE:382-423 (window_partition -> Attention -> window_unpartition), Attention.forward
// IE:465-501, add_decomposed_rel_pos IE:587-623, get_rel_pos IE:554-584.  Same math as attention.hip (which stays the
// kernel for global blocks, head_dim 32 and fp32 inputs); this one is shaped for the 20 windowed blocks of ViT-L:
//
//   * ONE workgroup per (window, head, image): 13 waves, wave w owns the 16 queries 16w .. 16w+15 of the 196 (<= 208).
//   * The window's whole K and V (all <= 196 keys, hi/lo planes) are brought into LDS ONCE by LDS-DMA straight from the
//     interleaved qkv planes -- a (token, 32 channels) unit is one 128-byte line in HBM and one 128-byte row in LDS
//     (the GEMM's LDS image and swizzle for K; V row-major with a 32-byte-unit swizzle for the transposed reads).
//     Window partition, the 64->70 padding and the un-partition are index arithmetic on the DMA source / output rows;
//     pad tokens read the qkv BIAS row (their k and v, IE:401-407,519-520) and are attended to like in the reference.
//   * One barrier.  After it every wave runs alone: S^T = K Q^T for all 13 key tiles (scores stay in 52 registers, so
//     the softmax is the exact two-pass form, no online rescaling), then O^T = V^T P^T.
//   * The rel-pos terms are no longer a separate pass: T[i][q] = rel_pos[i] . q for the 2*ws-1 relative offsets of each
//     axis is 4 more MFMA tiles per wave (unscaled q, split3 like everything else), re-indexed per query by key
//     coordinate (the integer gather of get_rel_pos, bit-exact) into a 32-entry row Bq[q] = [by kh | by kw]; the bias
//     then enters the score accumulators as one extra MFMA k-step per key tile against a constant 0/1 selector
//     sel[j] = onehot(kh(j)) | onehot(14 + kw(j)) -- the first version gathered two table entries per score element
//     with ~15 VALU instructions each and was VALU-bound (1500 VALU vs 186 MFMA per wave).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_AS __attribute__((address_space(3)))
#define GLDS16(gptr, lptr)                                                                                  \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                   \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

#define WA_WAVES 13
#ifndef WP_FOLD
#define WP_FOLD 1   // 0 (A/B builds): the scores are scaled in a pass of their own before the maxima (round 4)
#endif
#define WA_NKEY 208                      // key rows of the K image (13 tiles of 16)
#define WA_NKV 224                       // key rows of the V image (7 MFMA k-groups of 32)
#define WA_K_BYTES (2 * WA_NKEY * 128)   // [ks][key] rows of 128 B (4 hi chunks | 4 lo chunks of 32 channels)
#define WA_V_BYTES (WA_NKV * 128)        // per plane: [key] rows of 64 channels
#define WA_E_BYTES (WA_NKEY * 64)        // key -> (kh, 14 + kw) selector rows, 32 bf16 each
#define WA_B_BYTES (16 * 128)            // per wave: 16 queries x 32 floats (bias by kh | by kw)
#define WA_LDS (WA_K_BYTES + 2 * WA_V_BYTES + WA_E_BYTES + WA_WAVES * WA_B_BYTES)

struct WAttnArgs {
  const unsigned short* qp; long ldq;   // qkv planes [B*T, >= 2*3D]
  const unsigned short* bp;             // qkv bias planes [2*3D]
  const unsigned short* relp;           // rel-pos planes [64 rows, 2*64]: rows 0..2ws-2 = rel_pos_h, 32..32+2ws-2 = rel_pos_w
  const unsigned short* sel;            // [208, 32] bf16 selector: sel[j][kh(j)] = sel[j][14 + kw(j)] = 1 for j < ws*ws, else 0
  unsigned short* op; long ldo;         // output planes [B*T, >= 2*D]
  int ofmt;                             // their format (common.h: MMSA_FMT_B3 bf16 hi/lo, MMSA_FMT_H8 for an h8 proj GEMM)
  int B, H, W, heads, D, ws, nWw;
  float* guard;                         // optional device word: max |logit| (natural units, rel-pos bias included) over the live (query, key) pairs,
                                        // folded in with one atomic max per wave (include/mmsa.h "attention logit guard")
  unsigned magic;                       // ceil(65536 / ws): j / ws == (j * magic) >> 16 for j < 224 (checked on the host)
  float scale;
#ifdef MMSA_DEBUG_KNOBS
  long long* stamps;                    // debug builds only: shader-clock stamps of workgroup 0, waves 0 and 6, first 8 items x 16 slots
#endif
};
#ifdef MMSA_DEBUG_KNOBS   // tools/wattn_bench.py --stamps (tools/build_variant.sh ... -DMMSA_DEBUG_KNOBS); the release library has neither
static long long* g_wattn_stamps = nullptr;
extern "C" int mmsa_debug_wattn_stamps(long long* p) { g_wattn_stamps = p; return MMSA_OK; }
#define WP_STAMP(k_)                                                                                                         \
  do {                                                                                                                       \
    if (a.stamps && blockIdx.x == 0 && (wave == 0 || wave == 6) && lane == 0 && nst_ < 8)                                    \
      a.stamps[((wave ? 1 : 0) * 8 + nst_) * 16 + (k_)] = (long long)__builtin_amdgcn_s_memtime();                           \
  } while (0)
#else
#define WP_STAMP(k_) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// PERSISTENT kernel: one workgroup per CU walks the (window, head, image) items.  Measured on the round-1 one-shot kernel
// (one workgroup per item; tools/exp/r03_archive/wattn_variants.hip.txt): half of a workgroup's life was the cold start (launch, Q / table loads, the K/V transfer of
// 110 KiB that every CU requests at the same moment), during which nothing computes.  Here the next item's K is brought in
// while the current item's softmax and PV run (the K image is free once every wave has its scores), the V image while
// the next S runs, the next Q fragments are loaded into the registers the current ones vacate, and the rel-pos table and
// the selector are loaded once per workgroup.  Three barriers per item:
//   #1  K(i) landed for everyone, everyone is past PV(i-1)     -> V(i) DMA issued;   T / bias operand / S(i)
//   #2  everyone has its scores (K image free)                 -> K(i+1) DMA issued;   softmax statistics
//   #3  V(i) landed for everyone (counted vmcnt: K(i+1) stays in flight)             PV(i), store, Q(i+1) loads
#define WP_NKV 208
#define WP_V_BYTES (WP_NKV * 128)
#define WP_R_BYTES (2 * 64 * 128)        // rel-pos table image: [ks][64 rows] of 128 B (GEMM LDS image)
#define WP_LDS (WA_K_BYTES + 2 * WP_V_BYTES + WA_E_BYTES + WP_R_BYTES + WA_WAVES * WA_B_BYTES)

// VF: q, k, v, the pad-token bias row and the rel-pos table are h8 planes and every contraction of the kernel is ONE fp16 MFMA on
// their hi parts (attention.hip, attn_kernel VF = 2: same scheme, same error study): rel-pos terms 8 MFMAs instead of 24, Q K^T 2 per
// key tile instead of 6, P V 1 instead of 3 with P rounded to fp16; the rel-pos bias operand stays a hi + lo pair (two fp16 MFMAs
// against the fp16 selector).  The lo parts are not read; the V lo image is not transferred.
template <bool VF>
__global__ __launch_bounds__(WA_WAVES * 64) void wattn_persist_kernel(WAttnArgs a, int nWin, int nitems) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ks = smem;
  unsigned char* Vhi = smem + WA_K_BYTES;
  unsigned char* Vlo = Vhi + WP_V_BYTES;
  unsigned char* Es = Vlo + WP_V_BYTES;
  unsigned char* Rs = Es + WA_E_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Bw = reinterpret_cast<float*>(Rs + WP_R_BYTES + wave * WA_B_BYTES);
  const int l15 = lane & 15, G = lane >> 4;
  const int T = a.H * a.W, ws = a.ws, Nk = ws * ws;
  const int dr = lane >> 3, slot = lane & 7;

  // item -> (window row, window column, head, image); window-local index j -> token (-1 pad, -2 beyond the window)
  int wi, wj, head, b;
  // Items are ordered window-major with the windows that overhang the image (bottom row / right column: padded, fewer live
  // query tiles, cheaper) LAST: with a static round-robin the workgroups that get one item more than the others get these.
  const int nWh = nWin / a.nWw, nHB = a.heads * a.B;
  auto decode = [&](int it) {
    const int r = it / nHB;
    const int hb = it - r * nHB;
    head = hb % a.heads;
    b = hb / a.heads;
    const int nI = (nWh - 1) * (a.nWw - 1);
    if (r < nI) {
      wi = r / (a.nWw - 1);
      wj = r - wi * (a.nWw - 1);
    } else if (r - nI < nWh - 1) {
      wi = r - nI;
      wj = a.nWw - 1;
    } else {
      wi = nWh - 1;
      wj = r - nI - (nWh - 1);
    }
  };
  auto token_of = [&](int j) -> int {
    if (j >= Nk) return -2;
    const int r = (int)(__umul24(j, a.magic) >> 16), c = j - __umul24(r, ws);
    const int hh = wi * ws + r, ww = wj * ws + c;
    return (hh < a.H && ww < a.W) ? hh * a.W + ww : -1;
  };
  // (uses the CURRENT decode state)
#define WP_ISSUE_K()                                                                                          \
  {                                                                                                           \
    int lo_ = lane;   /* opaque: keep the per-lane address parts inside the item loop (register budget) */    \
    asm volatile("" : "+v"(lo_));                                                                             \
    const int dr = lo_ >> 3, slot = lo_ & 7;                                                                  \
    const unsigned short* pq_b_ = a.qp + (long)b * T * a.ldq;                                                 \
    const int colk_ = a.D + head * 64;                                                                        \
    _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                  \
      const int key = 16 * wave + 8 * half + dr;                                                              \
      const int t_ = token_of(key);                                                                           \
      const unsigned short* row = t_ >= 0 ? pq_b_ + (long)t_ * a.ldq : a.bp;                                  \
      const int piece = slot ^ ((key & 15) >> 1);                                                             \
      GLDS16(row + 2 * colk_ + piece * 8, Ks + (16 * wave + 8 * half) * 128);                                 \
      GLDS16(row + 2 * (colk_ + 32) + piece * 8, Ks + (WA_NKEY + 16 * wave + 8 * half) * 128);                \
    }                                                                                                         \
  }
#define WP_ISSUE_V()                                                                                          \
  {                                                                                                           \
    int lo_ = lane;                                                                                           \
    asm volatile("" : "+v"(lo_));                                                                             \
    const int dr = lo_ >> 3, slot = lo_ & 7;                                                                  \
    const unsigned short* pq_b_ = a.qp + (long)b * T * a.ldq;                                                 \
    const int colv_ = 2 * a.D + head * 64;                                                                    \
    _Pragma("unroll") for (int half = 0; half < 2; ++half) {                                                  \
      const int key = 16 * wave + 8 * half + dr;                                                              \
      const int t_ = token_of(key);                                                                           \
      const unsigned short* row = t_ >= 0 ? pq_b_ + (long)t_ * a.ldq : a.bp;                                  \
      const int c = slot ^ (((key >> 1) & 3) << 1);                                                           \
      const unsigned short* vsrc = row + 2 * (colv_ + 32 * (c >> 2)) + (c & 3) * 8;                           \
      GLDS16(vsrc, Vhi + (16 * wave + 8 * half) * 128);                                                       \
      if constexpr (!VF) GLDS16(vsrc + 32, Vlo + (16 * wave + 8 * half) * 128);                               \
    }                                                                                                         \
  }
#define WP_LOAD_Q()                                                                                           \
  {                                                                                                           \
    tq = token_of(16 * wave + l15);                                                                           \
    const unsigned short* qrow = a.qp + ((long)b * T + (tq >= 0 ? tq : 0)) * a.ldq;                           \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                        \
      const unsigned short* qq = qrow + 2 * (head * 64 + 32 * ks) + 8 * G;                                    \
      qh[ks] = *reinterpret_cast<const bf16x8*>(qq);                                                          \
      if constexpr (!VF) ql[ks] = *reinterpret_cast<const bf16x8*>(qq + 32);                                  \
    }                                                                                                         \
  }

  // ---- once per workgroup: selector tiles and the rel-pos table image
  {
    const int row = lane >> 2, ch = (lane & 3) ^ ((row >> 2) & 3);
    GLDS16(a.sel + (16 * wave + row) * 32 + ch * 8, Es + 16 * wave * 64);
#pragma unroll 1
    for (int u = wave; u < 16; u += WA_WAVES) {   // 16 groups of 8 rows: group u -> k-step u >> 3, rows 8*(u & 7) ..
      const int ks = u >> 3, r0 = 8 * (u & 7);
      const int rrow = r0 + dr;
      const int piece = slot ^ ((rrow & 15) >> 1);
      GLDS16(a.relp + rrow * 128 + 64 * ks + piece * 8, Rs + (ks * 64 + r0) * 128);
    }
  }
  int it = blockIdx.x;
  if (it >= nitems) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
  decode(it);
  bf16x8 qh[2], ql[2];
  int tq;
  WP_ISSUE_K()
  WP_LOAD_Q()

  const int fslot = G ^ ((l15 >> 1) & 7);
  const int frag_hi = l15 * 128 + fslot * 16, frag_lo = l15 * 128 + (fslot ^ 4) * 16;
  const int frag_e = l15 * 64 + ((G ^ ((l15 >> 2) & 3)) << 4);
  constexpr float LOG2E = 1.4426950408889634f;
  const float sc2 = a.scale * LOG2E, rscale = 1.0f / a.scale;
  bool first = true;
  float amax = 0.f;
  int nst_ = 0;   // items done (stamps of debug builds)

#pragma unroll 1
  for (;;) {
    const int jq = 16 * wave + l15;
    const bool live = tq >= 0;
    const bool any_live = __builtin_amdgcn_readfirstlane(__any(live) ? 1 : 0) != 0;   // wave-uniform
    const int tq_cur = tq;
    const int head_cur = head, b_cur = b;
    // rel-pos terms and the bias operand need only Q and the table image: done BEFORE barrier #1, in the time a wave
    // would otherwise wait for the slowest wave's PV / K transfer
    bf16x8 bqh = {0, 0, 0, 0, 0, 0, 0, 0}, bql = {0, 0, 0, 0, 0, 0, 0, 0};
    WP_STAMP(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q(i) (and this wave's K(i) DMA: older, so it has landed too)
    WP_STAMP(1);
    if (first) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); first = false; }   // table image of all waves
    if (any_live) {
      // rel-pos terms T[i][q] = rel_pos[i] . q (table fragments from the LDS image), re-indexed per query by key coordinate
      f32x4 tt[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        tt[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* rb = Rs + (ks * 64 + 16 * t) * 128;
          const bf16x8 rh_ = *reinterpret_cast<const bf16x8*>(rb + frag_hi);
          if constexpr (VF) {
            tt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, qh[ks]), tt[t], 0, 0, 0);
          } else {
            const bf16x8 rl_ = *reinterpret_cast<const bf16x8*>(rb + frag_lo);
            tt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rl_), __builtin_bit_cast(f16x8, qh[ks]), tt[t], 0, 0, 0);
            tt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, ql[ks]), tt[t], 0, 0, 0);
            tt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, qh[ks]), tt[t], 0, 0, 0);
          }
        }
      }
      {
        // lane indices through an opaque copy: LICM would otherwise hoist ~20 per-lane addresses of this block out of the
        // item loop and spill them (128-VGPR budget at 13 waves)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int l15 = lane_o & 15, G = lane_o >> 4;
        const int jqc = jq < Nk ? jq : 0;
        const int qr = (int)(__umul24(jqc, a.magic) >> 16), qc = jqc - __umul24(qr, ws);
        *reinterpret_cast<float4*>(Bw + l15 * 32 + 8 * G) = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Bw + l15 * 32 + 8 * G + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        // (round 5: the same writes made UNCONDITIONAL -- an offset outside the window sent to one of the row's four spare floats, which meet zeros of the
        // selector -- measured 45.5 -> 48.0 us per two-image launch, profiles/r05_wattn.txt: the exec-masked stores stay)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * (t & 1) + 4 * G + r;
            const int kc = ((t >> 1) ? qc : qr) + (ws - 1) - i;
            if (kc >= 0 && kc < ws) Bw[l15 * 32 + (t >> 1) * 14 + kc] = tt[t][r] * rscale;
          }
        const float4 b0 = *reinterpret_cast<const float4*>(Bw + l15 * 32 + 8 * G);
        const float4 b1 = *reinterpret_cast<const float4*>(Bw + l15 * 32 + 8 * G + 4);
        uint4 hh, ll;
        if constexpr (VF) {
          split2_f16(b0.x, b0.y, hh.x, ll.x);
          split2_f16(b0.z, b0.w, hh.y, ll.y);
          split2_f16(b1.x, b1.y, hh.z, ll.z);
          split2_f16(b1.z, b1.w, hh.w, ll.w);
        } else {
          split2_f16(b0.x, b0.y, hh.x, ll.x);
          split2_f16(b0.z, b0.w, hh.y, ll.y);
          split2_f16(b1.x, b1.y, hh.z, ll.z);
          split2_f16(b1.z, b1.w, hh.w, ll.w);
        }
        bqh = __builtin_bit_cast(bf16x8, hh);
        bql = __builtin_bit_cast(bf16x8, ll);
      }
    }
    // ---- barrier #1
    WP_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    WP_STAMP(3);
    WP_ISSUE_V()

    f32x4 s[13];
    if (any_live) {
      // S^T = sel Bq^T + K Q^T
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const bf16x8 e_ = *reinterpret_cast<const bf16x8*>(Es + 16 * t * 64 + frag_e);
        if constexpr (VF) {   // selector in fp16 (1.0 = 0x3C00), bias operand as an fp16 hi + lo pair
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, e_), __builtin_bit_cast(f16x8, bql), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, e_), __builtin_bit_cast(f16x8, bqh), s[t], 0, 0, 0);
        } else {
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, e_), __builtin_bit_cast(f16x8, bql), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, e_), __builtin_bit_cast(f16x8, bqh), s[t], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* kb = Ks + (ks * WA_NKEY + 16 * t) * 128;
          const bf16x8 kh_ = *reinterpret_cast<const bf16x8*>(kb + frag_hi);
          if constexpr (VF) {
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, qh[ks]), s[t], 0, 0, 0);
          } else {
            const bf16x8 kl_ = *reinterpret_cast<const bf16x8*>(kb + frag_lo);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kl_), __builtin_bit_cast(f16x8, qh[ks]), s[t], 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, ql[ks]), s[t], 0, 0, 0);
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, qh[ks]), s[t], 0, 0, 0);
          }
        }
        if (t & 1) __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting all 13 tiles' fragment reads (spills at 128 VGPRs)
      }
    }
    // ---- barrier #2: the K image and this wave's Q registers are free
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    WP_STAMP(4);
    __syncthreads();
    WP_STAMP(5);
    const int it_next = it + gridDim.x;
    const bool has_next = it_next < nitems;
    if (has_next) {
      decode(it_next);
      WP_ISSUE_K()
    }
    float mxs = 0.f;
    if (any_live) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int G = lane_o >> 4;
      // (round 5) the scores stay UNSCALED here: sc2 > 0, so the row maximum of the scaled scores is sc2 x this one, and the exponential below takes
      // fma(s, sc2, -sc2 * max) -- the 52 multiplies per lane of the scaling pass are gone; maxima and minima by threes (v_max3_f32 / v_min3_f32)
      float mx = -INFINITY, mn = INFINITY;
#pragma unroll
      for (int t = 0; t < 13; ++t) {
#if !WP_FOLD
        s[t] *= sc2;
#endif
        if (16 * t + 16 > Nk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ok = (16 * t + 4 * G + r) < Nk;
            mn = fminf(mn, ok ? s[t][r] : INFINITY);
            s[t][r] = ok ? s[t][r] : -INFINITY;
          }
        } else {
          mn = fminf(fminf(mn, s[t][0]), s[t][1]);
          mn = fminf(fminf(mn, s[t][2]), s[t][3]);
        }
        mx = fmaxf(fmaxf(mx, s[t][0]), s[t][1]);
        mx = fmaxf(fmaxf(mx, s[t][2]), s[t][3]);
      }
      if (live) amax = fmaxf(amax, (WP_FOLD ? sc2 : 1.0f) * fmaxf(fabsf(mx), fabsf(mn)));   // logit guard: this lane's query column, existing keys only
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      mxs = WP_FOLD ? mx * sc2 : mx;
    }
    // ---- barrier #3: V(i) landed (the 4 K DMA instructions of the next item, issued later, stay in flight)
    WP_STAMP(6);
    if (has_next) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WP_STAMP(7);
    __syncthreads();
    WP_STAMP(8);
    if (any_live) {
      int lane_o = lane;   // opaque copy: the 28 transposed-read addresses below must not be hoisted out of the item loop
      asm volatile("" : "+v"(lane_o));
      const int l15 = lane_o & 15, G = lane_o >> 4;
      f32x4 o[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) o[d] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float psum = 0.f;
#pragma unroll
      for (int g = 0; g < 7; ++g) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          if (2 * g + hf < 13) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = WP_FOLD ? __builtin_amdgcn_exp2f(fmaf(s[2 * g + hf][r], sc2, -mxs)) : __builtin_amdgcn_exp2f(s[2 * g + hf][r] - mxs);
              s[2 * g + hf][r] = p;
              psum += p;
            }
          }
        }
        uint4 hh, ll = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (VF) {
          hh.x = pack_f16(s[2 * g][0], s[2 * g][1]);
          hh.y = pack_f16(s[2 * g][2], s[2 * g][3]);
        } else {
          split2_f16(s[2 * g][0], s[2 * g][1], hh.x, ll.x);
          split2_f16(s[2 * g][2], s[2 * g][3], hh.y, ll.y);
        }
        if (2 * g + 1 < 13) {
          if constexpr (VF) {
            hh.z = pack_f16(s[2 * g + 1][0], s[2 * g + 1][1]);
            hh.w = pack_f16(s[2 * g + 1][2], s[2 * g + 1][3]);
          } else {
            split2_f16(s[2 * g + 1][0], s[2 * g + 1][1], hh.z, ll.z);
            split2_f16(s[2 * g + 1][2], s[2 * g + 1][3], hh.w, ll.w);
          }
        } else {
          hh.z = hh.w = ll.z = ll.w = 0u;   // keys 208..223 do not exist
        }
        const bf16x8 ph = __builtin_bit_cast(bf16x8, hh), pl = __builtin_bit_cast(bf16x8, ll);
        const int row0 = 32 * g + 4 * G + (l15 >> 2);
        const int sw = (row0 >> 1) & 3;
        const int second = (2 * g + 1 < 13) ? 16 * 128 : 0;   // no V rows beyond 207: re-read valid rows (their P is zero)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int voff = row0 * 128 + ((d ^ sw) << 5) + 8 * (l15 & 3);
          const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vhi + voff));
          const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vhi + voff + second));
          const bf16x8 vh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          if constexpr (VF) {
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, ph), o[d], 0, 0, 0);
          } else {
            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vlo + voff));
            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vlo + voff + second));
            const bf16x8 vl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vl), __builtin_bit_cast(f16x8, ph), o[d], 0, 0, 0);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, pl), o[d], 0, 0, 0);
            o[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, ph), o[d], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      psum += __shfl_xor(psum, 16, 64);
      psum += __shfl_xor(psum, 32, 64);
      const float inv = 1.0f / psum;
      {   // lanes G / G ^ 1 of a query row (lane ^ 16) hold 8 consecutive channels: whole-line stores through the pair exchange (common.h)
        const long orow = (long)b_cur * T + (tq_cur >= 0 ? tq_cur : 0);
#pragma unroll
        for (int d = 0; d < 4; ++d)
          store_planes8_pair_any<16>(a.op, a.ldo, orow, MMSA_PAD64(a.D), head_cur * 64 + 16 * d + 8 * (G >> 1), make_float4(o[d][0] * inv, o[d][1] * inv, o[d][2] * inv, o[d][3] * inv), a.ofmt, G & 1, tq_cur >= 0);
      }
    }
    WP_STAMP(9);
    ++nst_;
    if (!has_next) break;
    it = it_next;
    WP_LOAD_Q()   // (register pressure: loading them before PV spilled)
  }
  if (a.guard) {   // one atomic max per wave, and only when it would raise the word (after the first batches it never does)
    const float gv = wave_max(amax) * 0.6931471805599453f;
    if (lane == 0 && gv > *reinterpret_cast<volatile float*>(a.guard)) atomicMax(reinterpret_cast<unsigned*>(a.guard), __float_as_uint(gv));
  }
}

extern "C" int mmsa_window_attention_planes(const unsigned short* qkv_planes, long ldq, const unsigned short* bias_planes,
                                            const unsigned short* relpos_planes, const unsigned short* selector,
                                            unsigned short* out_planes, long ldo,
                                            int B, int H, int W, int heads, int head_dim, int window_size, float scale,
                                            int out_fmt, int v_fmt, float* max_abs_logit, hipStream_t stream) {
  MMSA_CHECK_ARG(qkv_planes && bias_planes && relpos_planes && selector && out_planes, "window_attention: null pointer");
  MMSA_CHECK_ARG(out_fmt >= MMSA_FMT_B3 && out_fmt <= MMSA_FMT_F3, "window_attention: bad output plane format %d", out_fmt);
  MMSA_CHECK_ARG(v_fmt == 0 || v_fmt == 2, "window_attention: v_fmt %d (0 = bf16 hi/lo planes; 2 = qkv, bias and rel-pos planes in the h8 format and an fp16 selector: every contraction on the fp16 MFMA)", v_fmt);
  MMSA_CHECK_ARG(B > 0 && H > 0 && W > 0 && heads > 0, "window_attention: bad shape");
  MMSA_CHECK_ARG(head_dim == 64, "window_attention: head_dim %d not supported by this kernel (64)", head_dim);
  MMSA_CHECK_ARG(window_size >= 1 && window_size <= 14, "window_attention: window_size %d not supported (1..14)", window_size);
  const int D = heads * head_dim;
  MMSA_CHECK_ARG(ldq >= 6L * D && (ldq & 63) == 0 && ldo >= (out_fmt == MMSA_FMT_H8C ? 3L : 2L) * D && (ldo & 63) == 0, "window_attention: bad leading dimensions");
  MMSA_CHECK_ARG(((reinterpret_cast<uintptr_t>(qkv_planes) | reinterpret_cast<uintptr_t>(bias_planes) |
                   reinterpret_cast<uintptr_t>(relpos_planes) | reinterpret_cast<uintptr_t>(selector) | reinterpret_cast<uintptr_t>(out_planes)) & 127) == 0,
                 "window_attention: planes must be 128-byte aligned");
  WAttnArgs a;
  a.qp = qkv_planes; a.ldq = ldq; a.bp = bias_planes; a.relp = relpos_planes; a.sel = selector; a.op = out_planes; a.ldo = ldo; a.ofmt = out_fmt;
  a.B = B; a.H = H; a.W = W; a.heads = heads; a.D = D; a.ws = window_size; a.scale = scale;
  a.nWw = cdiv(W, window_size);
  a.guard = max_abs_logit;
#ifdef MMSA_DEBUG_KNOBS
  a.stamps = g_wattn_stamps;
#endif
  a.magic = (unsigned)((65536 + window_size - 1) / window_size);
  for (int j = 0; j < WA_NKV; ++j)   // the kernel's multiply-shift division must be exact for every index it divides
    if ((int)(((unsigned)j * a.magic) >> 16) != j / window_size) {
      mmsa_set_error("window_attention: index arithmetic not exact for window_size %d", window_size);
      return MMSA_ERR_ARG;
    }
  // (cached per device: the launch attributes and the CU count of the current device -- common.h mmsa_per_device)
  static MmsaPerDevice per_dev_ = {};
  const int num_cus = mmsa_per_device(per_dev_, [] {
    (void)hipFuncSetAttribute((const void*)wattn_persist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
    (void)hipFuncSetAttribute((const void*)wattn_persist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS);
  });
  const int nWin = cdiv(H, window_size) * a.nWw;
  const int nitems = nWin * heads * B;
  // Grid: a workgroup walks items g, g + G, g + 2G, ... of a list that holds the interior windows first and the windows that
  // overhang the image (fewer live query tiles: cheaper) last.  Candidates: the fewest workgroups that finish in ceil(items / CUs)
  // items each (leaves CUs to concurrent streams) and one per CU; the cheaper schedule by a two-class cost model wins -- with 200
  // of 256 workgroups every workgroup of ViT-L's 25 x 16 x 2 items got 3 interior + 1 edge item, with 256 it is 2 + 1 (or 2).
  const int nWh_ = cdiv(H, window_size), nWw_ = a.nWw, nHB = heads * B;
  const int n_int = (nWh_ - 1) * (nWw_ - 1) * nHB;                    // items of interior windows (they come first)
  const int live_h = H - (nWh_ - 1) * window_size, live_w = W - (nWw_ - 1) * window_size;
  const double edge_cost = 0.35 + 0.65 * (0.5 * (live_h + live_w) / window_size);   // fixed part (K/V transfer, barriers) + live query tiles
  auto schedule_cost = [&](int G) {
    double worst = 0.0;
    for (int g = 0; g < G; ++g) {
      double c = 0.0;
      for (int it = g; it < nitems; it += G) c += it < n_int ? 1.0 : edge_cost;
      worst = c > worst ? c : worst;
    }
    return worst;
  };
  const int rounds = cdiv(nitems, num_cus);
  int grid = cdiv(nitems, rounds);
  const int grid_all = nitems < num_cus ? nitems : num_cus;
  if (schedule_cost(grid_all) < schedule_cost(grid) - 1e-9) grid = grid_all;
  if (v_fmt) hipLaunchKernelGGL(wattn_persist_kernel<true>, dim3(grid), dim3(WA_WAVES * 64), WP_LDS, stream, a, nWin, nitems);
  else hipLaunchKernelGGL(wattn_persist_kernel<false>, dim3(grid), dim3(WA_WAVES * 64), WP_LDS, stream, a, nWin, nitems);
  MMSA_CHECK_LAUNCH("window_attention");
  return MMSA_OK;
}
