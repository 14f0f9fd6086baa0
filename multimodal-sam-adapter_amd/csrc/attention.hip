// SAM ViT attention with decomposed relative position bias, for gfx950.
//
// Reference: Attention.forward IE:465-501 (scores = (q*scale) k^T + rel_h[q,kh] + rel_w[q,kw], softmax over ALL
// keys incl. the zero-padded window tokens, then @v), window_partition / window_unpartition IE:504-551 and
// add_decomposed_rel_pos IE:587-623 (bias from the UNSCALED q).
//
// One flash-style kernel serves both block kinds:
//   * global blocks (window_size 0): one key set of H*W tokens per image (the reference materialises the
//     [heads, 4096, 4096] score tensor; here it never leaves registers);
//   * windowed blocks: key set = the ws*ws tokens of one window.  Window partition, the 64->70 zero padding,
//     unpartition and the crop are pure index arithmetic in the load/store paths (no copies).  Pad tokens are
//     zeros AFTER norm1, so their k and v equal the qkv bias (IE:401-407,519-520): they are synthesised from the
//     bias vector instead of being run through the qkv GEMM, and they ARE attended to (no mask), like the reference.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16, split3 operands hi/lo):  S^T = K Q^T is computed "swapped" so a lane
// owns one query column: softmax statistics are per-lane scalars (+2 shuffles across the 4 lane groups), and the
// exponentiated tile is directly the B operand of O^T = V^T P^T (no LDS round trip, no transposes of P).  V^T
// fragments come from row-major V in LDS through ds_read_b64_tr_b16.  A wave owns 32 queries (2 sub-tiles), a
// workgroup 128; keys stream through LDS in blocks of 64 with register prefetch of the next block.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;
#define LDS_AS __attribute__((address_space(3)))

struct AttnArgs {
  const float* qkv; long ldq;    // [B*T, 3*D]: q | k | v, channel = head*HD + c   (IE:488 memory order)
  const float* qkv_bias;         // [3*D]
  const unsigned short* qp;      // ilv planes form of qkv (PL kernels), row stride ldq (bf16 units, >= 2*3D)
  const unsigned short* bp;      // ilv planes form of qkv_bias [2*3D]
  unsigned short* op;            // ilv planes output (PL kernels), row stride ldo (>= 2*D)
  int ofmt;                      // format of the planes output (common.h): bf16 hi/lo, or h8 for an h8 proj GEMM
  int vf;                        // PL kernels: 1 = the v columns of qp / bp are h8 planes (fp16 hi): P V on the fp16 MFMA (VF kernels)
  const float* rp;               // [B, heads, T, KH+KW] rel-pos bias terms (relpos kernel below)
  const unsigned short* relg;    // REL kernels: planes of a [256, 64] matrix, rows 0..2KH-2 = rel_pos_h, rows 128..128+2KW-2 = rel_pos_w
  float* out; long ldo;          // [B*T, D], channel = head*HD + c (IE:498)
  int B, H, W, heads, D;
  int ws;                        // 0: global attention; >0: window size
  int nWw;                       // windows per row (ceil(W/ws))
  int Nk;                        // keys (= queries) per group: H*W or ws*ws
  int KH, KW;                    // bias table extents (H,W) or (ws,ws)
  int KHs, KWs;                  // odd LDS row strides for the bias tables
  unsigned magicKW;              // ceil(2^24 / KW): j / KW == (j * magic) >> 24 for the j used here
  float scale;
  float* guard;                  // optional device word: max |logit| (natural units, rel-pos terms included) over every query / key pair the launch
                                 // scores is folded into it with one atomic max per wave (include/mmsa.h "attention logit guard")
};

// REL (with PL, FB, HD = 64): the rel-pos terms are computed in the kernel's prologue (MFMA, like wattn.hip) instead of being
// read from the prepass output
// VF = 2 (with REL): q, k, v, the pad-token bias row and the rel-pos tables are h8 planes and EVERY contraction of the kernel is one
// fp16 MFMA on their hi parts -- rel-pos terms, Q K^T (two MFMAs per score tile instead of six) and P V; the lo parts are neither
// loaded nor staged.  Same study: 3.2e-5 relative on f1..f4 with every operand of every attention block rounded to fp16 (the
// rounding errors of q and k are random and average out over the head dimension and the keys; scores are O(10)).
// VF = 1 (with PL): the v columns of the qkv planes carry an fp16 hi part (h8 planes: the qkv GEMM writes them that way, common.h
// MMSA_CP_SPLIT) and P V runs as ONE fp16 MFMA per product with P rounded to fp16 -- instead of three bf16 MFMAs on hi/lo pairs of
// both.  P <= 1 has 11 significant bits in fp16 and its rounding errors average out over the keys; measured on the CPU oracle
// (ViT-B 512^2, every attention block): 2.9e-5 relative on f1..f4 against 1.2e-4 for a bf16 P (LAB_NOTES.md 4.1).  Q K^T stays on
// bf16 hi/lo pairs: the scores are exponentiated.
template <int HD, bool PL, bool FB, bool REL = false, int VF = 0>
__global__ __launch_bounds__(256, (FB && HD <= 64) ? 2 : 1) void attn_kernel(AttnArgs a) {
  static_assert(VF == 0 || PL, "the fp16 paths read planes");
  static_assert(VF != 2 || REL, "the all-fp16 form exists for the fused rel-pos kernel");
  constexpr int NCH = HD / 8;            // 16-byte k-chunks per row
  constexpr int KS = HD / 32;            // MFMA k-steps over the head dim
  constexpr int DT = HD / 16;            // output d tiles
  constexpr int VSTR = 2 * HD + 32;      // V row stride in bytes: 96 / 160 / 224 -- 8 consecutive rows tile the 64 banks exactly (conflict-free tr reads)
  constexpr int KPL = NCH * 64 * 16;     // bytes per K plane
  constexpr int VPL = 64 * VSTR;         // bytes per V plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Khi = smem;
  unsigned char* Klo = Khi + KPL;
  unsigned char* Vhi = Klo + KPL;
  unsigned char* Vlo = Vhi + VPL;
  float* bh = reinterpret_cast<float*>(Vlo + VPL);
  float* bw = bh + 128 * a.KHs;   // FB: not allocated (the W-term goes straight from HBM to registers)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, G = lane >> 4;
  // XCD-contiguous order (common.h): the query blocks of one (image, head) -- which all read that head's K and V -- run on ONE XCD, so its L2 holds
  // them; in dispatch order they went to all eight (614 MB read per launch for 60 MB of q, k, v: profiles/r04_xcd_order.txt)
  const unsigned wg_ = mmsa_xcd_order(mmsa_block_lin(), mmsa_block_count());
  const int bx_ = (int)(wg_ % gridDim.x);
  const int head = (int)((wg_ / gridDim.x) % gridDim.y), b = (int)(wg_ / (gridDim.x * gridDim.y));
  const int T = a.H * a.W;
  const int nqb = (a.Nk + 127) / 128;
  const int grp = bx_ / nqb;
  const int q0 = (bx_ % nqb) * 128;
  const int wi = a.ws ? grp / a.nWw : 0, wj = a.ws ? grp % a.nWw : 0;

  // group index j -> token (or -1 for a zero-pad token, -2 for "does not exist")
  auto token_of = [&](int j) -> int {
    if (j >= a.Nk) return -2;
    if (a.ws == 0) return j;
    const int r = j / a.ws, c = j - r * a.ws;
    const int hh = wi * a.ws + r, ww = wj * a.ws + c;
    return (hh < a.H && ww < a.W) ? hh * a.W + ww : -1;
  };

  const float* qkv_b = PL ? nullptr : a.qkv + (long)b * T * a.ldq + head * HD;
  const float* kbias = PL ? nullptr : a.qkv_bias + a.D + head * HD;
  const float* vbias = PL ? nullptr : a.qkv_bias + 2 * a.D + head * HD;
  const unsigned short* pq_b = PL ? a.qp + (long)b * T * a.ldq : nullptr;   // image base of the ilv planes
  const int colq = head * HD, colk = a.D + head * HD, colv = 2 * a.D + head * HD;

  // ---- bias tables for the block's 128 queries -> LDS
  if constexpr (REL) {
    // filled below from the Q fragments
  } else if constexpr (FB) {
    // H-term only (KH % 4 == 0, checked by the launcher): float4 loads, four per lane IN FLIGHT before the first LDS
    // write -- the rolled load -> store loop below serialises one global round trip per element
    const int ncol = a.KH + a.KW, nf4 = a.KH >> 2, total4 = 128 * nf4;
    const float* rpb = a.rp + ((long)b * a.heads + head) * T * ncol;
    for (int i0 = tid; i0 < total4; i0 += 1024) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        const int ql = i / nf4, c4 = i - ql * nf4;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total4 && q0 + ql < a.Nk) v[u] = *reinterpret_cast<const float4*>(rpb + (long)(q0 + ql) * ncol + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 256 * u;
        if (i < total4) {
          const int ql = i / nf4, c4 = i - ql * nf4;
          float* d = bh + ql * a.KHs + 4 * c4;
          d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
        }
      }
    }
  } else {
    const int ncol = a.KH + a.KW;
    const float* rpb = a.rp + ((long)b * a.heads + head) * T * ncol;
    for (int i = tid; i < 128 * ncol; i += 256) {
      const int ql = i / ncol, cidx = i - ql * ncol;
      const int tq = token_of(q0 + ql);
      const float v = tq >= 0 ? rpb[(long)tq * ncol + cidx] : 0.f;
      if (cidx < a.KH) bh[ql * a.KHs + cidx] = v;
      else bw[ql * a.KWs + (cidx - a.KH)] = v;
    }
  }

  // ---- Q fragments (B operand: B[k = 8G + j][col = query l15]) for the wave's two 16-query sub-tiles
  bf16x8 qh[2][KS], ql_[2][KS];
  int tq_sub[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int jq = q0 + wave * 32 + sub * 16 + l15;
    const int tq = token_of(jq);
    tq_sub[sub] = tq;
    const long qrow = (long)(tq >= 0 ? tq : 0) * a.ldq;
    const float* qp = qkv_b + qrow;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (PL) {  // planes: fragments are plain 16-byte loads; the softmax scale is applied to S instead
        const unsigned short* qq = pq_b + qrow + ilv(colq + ks * 32 + 8 * G);
        qh[sub][ks] = *reinterpret_cast<const bf16x8*>(qq);
        if constexpr (VF != 2) ql_[sub][ks] = *reinterpret_cast<const bf16x8*>(qq + 32);
        continue;
      }
      float4 v0 = *reinterpret_cast<const float4*>(qp + ks * 32 + 8 * G);
      float4 v1 = *reinterpret_cast<const float4*>(qp + ks * 32 + 8 * G + 4);
      v0.x *= a.scale; v0.y *= a.scale; v0.z *= a.scale; v0.w *= a.scale;
      v1.x *= a.scale; v1.y *= a.scale; v1.z *= a.scale; v1.w *= a.scale;
      uint2 h0, l0, h1, l1;
      f3_split4(v0, h0, l0);
      f3_split4(v1, h1, l1);
      uint4 hh4 = make_uint4(h0.x, h0.y, h1.x, h1.y), ll4 = make_uint4(l0.x, l0.y, l1.x, l1.y);
      qh[sub][ks] = __builtin_bit_cast(bf16x8, hh4);
      ql_[sub][ks] = __builtin_bit_cast(bf16x8, ll4);
    }
  }

  // FB: W-term of the bias for this lane's fixed key columns (kw = 16t + 4G + r), pre-multiplied by log2(e)
  float bwr[2][4][4];
  if constexpr (REL) {
    // T[i][q] = rel_pos[i] . q (unscaled q) for the 2K-1 relative offsets of each axis: 8 MFMA tiles per axis and sub-tile,
    // table fragments straight from the packed planes (64 KiB, L2-resident).  Re-indexed by key coordinate
    // (get_rel_pos IE:579-584: i = (q - k) + (K - 1)): H-term -> bh[q][kh] (LDS, read once per key block), W-term -> a
    // temporary table in the K/V staging region, from which each lane takes the 16 values of its fixed key columns.
    float* bwt = reinterpret_cast<float*>(smem);
    int qc[2][2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int tok = min(q0 + wave * 32 + sub * 16 + l15, a.Nk - 1);
      qc[sub][0] = tok / a.W;
      qc[sub][1] = tok - qc[sub][0] * a.W;
    }
#pragma unroll
    for (int ax = 0; ax < 2; ++ax) {
      const int Kx = ax ? a.KW : a.KH;
      float* tab = ax ? bwt : bh;
      const int tstr = ax ? a.KWs : a.KHs;
#pragma unroll 2
      for (int t = 0; t < 8; ++t) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const unsigned short* rr = a.relg + (long)(ax * 128 + 16 * t + l15) * 128 + 64 * ks + 8 * G;
          const bf16x8 rh_ = *reinterpret_cast<const bf16x8*>(rr);
          if constexpr (VF == 2) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
              acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, qh[sub][ks]), acc[sub], 0, 0, 0);
          } else {
            const bf16x8 rl_ = *reinterpret_cast<const bf16x8*>(rr + 32);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
              acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rl_), __builtin_bit_cast(f16x8, qh[sub][ks]), acc[sub], 0, 0, 0);
              acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, ql_[sub][ks]), acc[sub], 0, 0, 0);
              acc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, rh_), __builtin_bit_cast(f16x8, qh[sub][ks]), acc[sub], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kc = qc[sub][ax] + (Kx - 1) - (16 * t + 4 * G + r);   // key coordinate served by table row i
            if (kc >= 0 && kc < Kx) tab[(wave * 32 + sub * 16 + l15) * tstr + kc] = acc[sub][r];
          }
      }
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          bwr[sub][t][r] = bwt[(wave * 32 + sub * 16 + l15) * a.KWs + 16 * t + 4 * G + r] * 1.4426950408889634f;
    __syncthreads();   // the K/V staging below overwrites the temporary table
  } else if constexpr (FB) {
    // straight from the rel-pos prepass output (no LDS copy: the table is read exactly once per lane), which leaves
    // 36 KiB K/V + 33 KiB H-term table per workgroup = two workgroups per CU
    const float* rpb = a.rp + ((long)b * a.heads + head) * T * (a.KH + a.KW) + a.KH;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int tq = tq_sub[sub];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tq >= 0) v = *reinterpret_cast<const float4*>(rpb + (long)tq * (a.KH + a.KW) + 16 * t + 4 * G);
        bwr[sub][t][0] = v.x * 1.4426950408889634f;
        bwr[sub][t][1] = v.y * 1.4426950408889634f;
        bwr[sub][t][2] = v.z * 1.4426950408889634f;
        bwr[sub][t][3] = v.w * 1.4426950408889634f;
      }
    }
  }

  // ---- K/V staging: thread -> (key, quarter of the head dim), HD/4 channels of K and of V.  Lanes 0-7 of every
  // 8-lane ds_write_b128 group take 8 consecutive keys of ONE quarter, so a group writes 128 contiguous bytes of
  // the K image (key = tid/4, quarter = tid%4 would put 4 lanes of a group on the same banks: 4-way conflict).
  constexpr int NF4 = HD / 16;  // float4 per thread per operand (fp32 input)
  constexpr int NU = HD / 32;   // uint4 per thread per plane per operand (planes input)
  const int skey = (tid & 7) + 8 * (tid >> 5), squart = (tid >> 3) & 3;
  float4 rk[NF4], rv[NF4];
  uint4 rkh[NU], rkl[NU], rvh[NU], rvl[NU];
  unsigned kv_keep = 0xffffffffu;   // 0 for a key beyond Nk: its staged registers hold a valid row's bytes and are zeroed on the way into LDS
  auto and4 = [](uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); };
#define LOAD_KV(kb_)                                                                         \
  do {                                                                                       \
    const int tk_ = token_of((kb_) * 64 + skey);                                             \
    if constexpr (PL) {                                                                      \
      /* ONE unconditional load per register: a token row, or the qkv bias row for a pad token (tk_ == -1) and -- as a valid   \
         address only -- for a key that does not exist (tk_ == -2: zeroed afterwards).  The three-way `if` around the loads      \
         compiled to exec-masked loads into the same registers with a full vmcnt wait between them (write-after-write): two       \
         exposed memory round trips per key block, ~3 of the 4.6 us a block took */                                            \
      const unsigned short* rowp_ = tk_ >= 0 ? pq_b + (long)tk_ * a.ldq : a.bp;                                                \
      const unsigned keep_ = tk_ == -2 ? 0u : 0xffffffffu;                                                                     \
      _Pragma("unroll") for (int i = 0; i < NU; ++i) {                                       \
        const int c = squart * (HD / 4) + 8 * i;                                             \
        const unsigned short* kr_ = rowp_ + ilv(colk + c);                                   \
        const unsigned short* vr_ = rowp_ + ilv(colv + c);                                   \
        rkh[i] = *reinterpret_cast<const uint4*>(kr_);                                       \
        if constexpr (VF != 2) rkl[i] = *reinterpret_cast<const uint4*>(kr_ + 32);           \
        rvh[i] = *reinterpret_cast<const uint4*>(vr_);                                       \
        if constexpr (VF == 0) rvl[i] = *reinterpret_cast<const uint4*>(vr_ + 32);           \
      }                                                                                      \
      kv_keep = keep_;                                                                       \
    } else {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < NF4; ++i) {                                      \
        const int c = squart * (HD / 4) + 4 * i;                                             \
        if (tk_ >= 0) {                                                                      \
          const float* p = qkv_b + (long)tk_ * a.ldq + c;                                    \
          rk[i] = *reinterpret_cast<const float4*>(p + a.D);                                 \
          rv[i] = *reinterpret_cast<const float4*>(p + 2 * a.D);                             \
        } else if (tk_ == -1) {                                                              \
          rk[i] = *reinterpret_cast<const float4*>(kbias + c);                               \
          rv[i] = *reinterpret_cast<const float4*>(vbias + c);                               \
        } else {                                                                             \
          rk[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                           \
          rv[i] = rk[i];                                                                     \
        }                                                                                    \
      }                                                                                      \
    }                                                                                        \
  } while (0)
#define STORE_KV(Kh_, Kl_, Vh_, Vl_)                                                                           \
  do {                                                                                       \
    if constexpr (PL) {                                                                      \
      _Pragma("unroll") for (int i = 0; i < NU; ++i) {                                       \
        const int c = squart * (HD / 4) + 8 * i;                                             \
        const int ko = (c >> 3) * (64 * 16) + skey * 16;                                     \
        *reinterpret_cast<uint4*>((Kh_) + ko) = and4(rkh[i], kv_keep);                         \
        if constexpr (VF != 2) *reinterpret_cast<uint4*>((Kl_) + ko) = and4(rkl[i], kv_keep);  \
        const int vo = skey * VSTR + c * 2;                                                  \
        *reinterpret_cast<uint4*>((Vh_) + vo) = and4(rvh[i], kv_keep);                         \
        if constexpr (VF == 0) *reinterpret_cast<uint4*>((Vl_) + vo) = and4(rvl[i], kv_keep);  \
      }                                                                                      \
    } else {                                                                                 \
      _Pragma("unroll") for (int i = 0; i < NF4; ++i) {                                      \
        const int c = squart * (HD / 4) + 4 * i;                                             \
        uint2 h, l;                                                                          \
        f3_split4(rk[i], h, l);                                                              \
        const int ko = (c >> 3) * (64 * 16) + skey * 16 + (c & 7) * 2;                       \
        *reinterpret_cast<uint2*>((Kh_) + ko) = h;                                             \
        *reinterpret_cast<uint2*>((Kl_) + ko) = l;                                             \
        f3_split4(rv[i], h, l);                                                              \
        const int vo = skey * VSTR + c * 2;                                                  \
        *reinterpret_cast<uint2*>((Vh_) + vo) = h;                                             \
        *reinterpret_cast<uint2*>((Vl_) + vo) = l;                                             \
      }                                                                                      \
    }                                                                                        \
  } while (0)

  f32x4 o[2][DT];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < DT; ++d) o[s][d] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {-INFINITY, -INFINITY};
  float l_run[2] = {0.f, 0.f};
  float amax = 0.f;   // logit guard: largest |score| (log2 units) this lane's LIVE queries have seen

  const int nkb = (a.Nk + 63) / 64;
  // VF = 2, MMSA_ATTN_DB = 1 (A/B builds; round 5): the lo planes' LDS regions are unused by the all-fp16 form and can serve as a SECOND K / V buffer, so that a key
  // block costs one barrier instead of two: block kb + 1 is written (from the registers its loads filled during block kb's arithmetic) into the buffer block kb - 1 was
  // read from, which every wave left before the barrier that ended block kb - 1.  Measured no faster (257.5 -> 261 us per two-image launch, profiles/r05_attention.txt):
  // with two workgroups per CU the barriers were not what a key block waits for -- its softmax is issue time (profiles/r05_gelu_forms.txt has the issue costs).  Off.
#ifndef MMSA_ATTN_DB
#define MMSA_ATTN_DB 0
#endif
  constexpr bool DB = (VF == 2) && MMSA_ATTN_DB;
  LOAD_KV(0);
  if constexpr (DB) {
    __syncthreads();   // (orders the bias-table fill; the staging region is free)
    STORE_KV(Khi, Klo, Vhi, Vlo);
    __syncthreads();
    if (nkb > 1) LOAD_KV(1);
  }
  for (int kb = 0; kb < nkb; ++kb) {
    if constexpr (!DB) {
      __syncthreads();  // previous block fully consumed (also orders the bias-table fill on kb == 0)
      STORE_KV(Khi, Klo, Vhi, Vlo);
      __syncthreads();
      if (kb + 1 < nkb) LOAD_KV(kb + 1);
    }
    const unsigned char* Krd = (DB && (kb & 1)) ? Klo : Khi;   // the buffer this block is read from
    const unsigned char* Vrd = (DB && (kb & 1)) ? Vlo : Vhi;

    // ---- S^T tiles: [t = key tile][sub]; lane holds keys 16t + 4G + r (r = reg), query column l15
    f32x4 s[2][4];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int t = 0; t < 4; ++t) s[sub][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (ks * 4 + G) * (64 * 16) + (16 * t + l15) * 16;
        const bf16x8 kh_ = *reinterpret_cast<const bf16x8*>(Krd + off);
        if constexpr (VF == 2) {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
            s[sub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, qh[sub][ks]), s[sub][t], 0, 0, 0);
        } else {
          const bf16x8 kl_ = *reinterpret_cast<const bf16x8*>(Klo + off);
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            s[sub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kl_), __builtin_bit_cast(f16x8, qh[sub][ks]), s[sub][t], 0, 0, 0);
            s[sub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, ql_[sub][ks]), s[sub][t], 0, 0, 0);
            s[sub][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, kh_), __builtin_bit_cast(f16x8, qh[sub][ks]), s[sub][t], 0, 0, 0);
          }
        }
      }
    }

    // ---- bias, mask, online softmax -- all in log2 units (scores and bias pre-multiplied by log2(e)) so that the
    // exponential is a bare v_exp_f32
    bf16x8 ph[2][2], pl[2][2];  // [sub][k-step of 32 keys]
    constexpr float LOG2E = 1.4426950408889634f;
    const float sc2 = (PL ? a.scale : 1.0f) * LOG2E;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int qloc = wave * 32 + sub * 16 + l15;
      const float* bhq = bh + qloc * a.KHs;
      const float* bwq = bw + qloc * a.KWs;
      float mx = -INFINITY;
      float bhv_blk = 0.f;   // FB: H-term of this key block (scores below exclude it)
      if constexpr (FB) {
        // global attention on a 64-wide grid: the 64 keys of a block are one image row (kh = kb) and
        // kw = 16t + 4G + r is the same in every block -> the W-term lives in registers, the H-term is one read
        // the H-term is the same for all 64 keys of the block: it is added to the row maximum, not to every score
        bhv_blk = bhq[kb] * LOG2E;
        float mn = INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const float v0 = s[sub][t][r] * sc2 + bwr[sub][t][r], v1 = s[sub][t][r + 1] * sc2 + bwr[sub][t][r + 1];
            s[sub][t][r] = v0;
            s[sub][t][r + 1] = v1;
            mx = fmaxf(fmaxf(mx, v0), v1);     // v_max3_f32 / v_min3_f32: half an instruction per score for the guard
            mn = fminf(fminf(mn, v0), v1);
          }
        mx += bhv_blk;
        if (tq_sub[sub] >= 0) amax = fmaxf(fmaxf(amax, fabsf(mx)), fabsf(mn + bhv_blk));
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = kb * 64 + 16 * t + 4 * G + r;
            float v;
            if (j < a.Nk) {
              const int kh = (int)(((unsigned)j * a.magicKW) >> 24);
              const int kw = j - kh * a.KW;
              v = s[sub][t][r] * sc2 + (bhq[kh] + bwq[kw]) * LOG2E;
              if (tq_sub[sub] >= 0) amax = fmaxf(amax, fabsf(v));
            } else {
              v = -INFINITY;
            }
            s[sub][t][r] = v;
            mx = fmaxf(mx, v);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[sub], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[sub] - m_new);
      // wave-uniform: no row of this sub-tile has a new maximum -> nothing to rescale (common after the first key blocks)
      const bool rescale = __builtin_amdgcn_readfirstlane(__any(m_new != m_run[sub]) ? 1 : 0) != 0;
      m_run[sub] = m_new;
      const float m_sub = m_new - bhv_blk;   // scores exclude the block's H-term
      float psum = 0.f;
      // MFMA k-slot (G, j): j < 4 -> key 16*(2*s2) + 4G + j ; j >= 4 -> key 16*(2*s2+1) + 4G + (j-4)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        uint4 hh, ll;
        unsigned* hp = reinterpret_cast<unsigned*>(&hh);
        unsigned* lp = reinterpret_cast<unsigned*>(&ll);
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // dword u = elements (2u, 2u+1): tile 2*s2 + (u>>1), regs 2*(u&1), 2*(u&1)+1
          const int t = 2 * s2 + (u >> 1), r = 2 * (u & 1);
          const float p0 = __builtin_amdgcn_exp2f(s[sub][t][r] - m_sub);
          const float p1 = __builtin_amdgcn_exp2f(s[sub][t][r + 1] - m_sub);
          psum += p0 + p1;
          if constexpr (VF) { hp[u] = pack_f16(p0, p1); lp[u] = 0u; }
          else split2_f16(p0, p1, hp[u], lp[u]);
        }
        ph[sub][s2] = __builtin_bit_cast(bf16x8, hh);
        pl[sub][s2] = __builtin_bit_cast(bf16x8, ll);
      }
      l_run[sub] = l_run[sub] * alpha + psum;
      if (rescale) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          o[sub][d][0] *= alpha; o[sub][d][1] *= alpha; o[sub][d][2] *= alpha; o[sub][d][3] *= alpha;
        }
      }
    }

    // ---- O^T += V^T P^T
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        // lane i = 4q'+p of a 16-lane group addresses row q' (key), columns 4p..4p+3 of the 4x16 block
        const int row0 = 32 * s2 + 4 * G + (l15 >> 2);
        const int voff = row0 * VSTR + (16 * d + 4 * (l15 & 3)) * 2;
        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vrd + voff));
        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vrd + voff + 16 * VSTR));
        const bf16x8 vh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        if constexpr (VF) {
#pragma unroll
          for (int sub = 0; sub < 2; ++sub)
            o[sub][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, ph[sub][s2]), o[sub][d], 0, 0, 0);
        } else {
          const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vlo + voff));
          const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS s16x4*)(Vlo + voff + 16 * VSTR));
          const bf16x8 vl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            o[sub][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vl), __builtin_bit_cast(f16x8, ph[sub][s2]), o[sub][d], 0, 0, 0);
            o[sub][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, pl[sub][s2]), o[sub][d], 0, 0, 0);
            o[sub][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, vh), __builtin_bit_cast(f16x8, ph[sub][s2]), o[sub][d], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (DB) {
      if (kb + 1 < nkb) {   // (uniform) block kb + 1 into the other buffer, then ONE barrier: it is visible, and this block's buffer is free for block kb + 2
        if (kb & 1) STORE_KV(Khi, Klo, Vhi, Vlo); else STORE_KV(Klo, Klo, Vlo, Vlo);
        __syncthreads();
        if (kb + 2 < nkb) LOAD_KV(kb + 2);
      }
    }
  }

  // ---- logit guard: one atomic max per wave, and only when it would raise the word (after the first batches it never does)
  if (a.guard) {
    const float gv = wave_max(amax) * 0.6931471805599453f;
    if (lane == 0 && gv > *reinterpret_cast<volatile float*>(a.guard)) atomicMax(reinterpret_cast<unsigned*>(a.guard), __float_as_uint(gv));
  }
  // ---- epilogue: O[q][16d + 4G .. +3] / l  -> out[token(q)][head*HD + ...]
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float l = l_run[sub];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int tq = tq_sub[sub];
    if constexpr (PL) {
      // lanes G / G ^ 1 of a query row (lane ^ 16) hold 8 consecutive channels: whole-line stores through the pair exchange (common.h)
      const long orow = (long)b * T + (tq >= 0 ? tq : 0);
#pragma unroll
      for (int d = 0; d < DT; ++d)
        store_planes8_pair_any<16>(a.op, a.ldo, orow, MMSA_PAD64(a.D), head * HD + 16 * d + 8 * (G >> 1),
                                   make_float4(o[sub][d][0] * inv, o[sub][d][1] * inv, o[sub][d][2] * inv, o[sub][d][3] * inv), a.ofmt, G & 1, tq >= 0);
    } else if (tq >= 0) {
      const long oo = ((long)b * T + tq) * a.ldo + head * HD + 4 * G;
#pragma unroll
      for (int d = 0; d < DT; ++d)
        *reinterpret_cast<float4*>(a.out + oo + 16 * d) = make_float4(o[sub][d][0] * inv, o[sub][d][1] * inv, o[sub][d][2] * inv, o[sub][d][3] * inv);
    }
  }
}

static int attention_launch(AttnArgs a, int B, int H, int W, int heads, int head_dim, int window_size, float scale,
                            bool planes, hipStream_t stream) {
  MMSA_CHECK_ARG(B > 0 && H > 0 && W > 0 && heads > 0 && window_size >= 0, "attention: bad shape");
  MMSA_CHECK_ARG(head_dim == 64 || head_dim == 32 || head_dim == 96, "attention: head_dim %d not supported (32, 64 or 96; other widths are zero-padded per head by the caller, e.g. ViT-H's 80 -> 96)", head_dim);
  const int D = heads * head_dim;
  MMSA_CHECK_ARG(a.ldq >= (planes ? 6L : 3L) * D && (a.ldq & 7) == 0 && a.ldo >= (planes ? (a.ofmt == MMSA_FMT_H8C ? 3L * MMSA_PAD64(D) : 2L * D) : 1L * D) && (a.ldo & 3) == 0,
                 "attention: bad leading dimensions");
  a.B = B; a.H = H; a.W = W; a.heads = heads; a.D = D; a.ws = window_size; a.scale = scale;
  int ngroups;
  if (window_size > 0) {
    a.nWw = cdiv(W, window_size);
    ngroups = cdiv(H, window_size) * a.nWw;
    a.Nk = window_size * window_size;
    a.KH = a.KW = window_size;
  } else {
    a.nWw = 1;
    ngroups = 1;
    a.Nk = H * W;
    a.KH = H;
    a.KW = W;
  }
  a.KHs = a.KH | 1;
  a.KWs = a.KW | 1;
  a.magicKW = (unsigned)(((1u << 24) + a.KW - 1) / a.KW);
  for (int j = 0; j < a.Nk; ++j) {  // the kernel's multiply-shift division must be exact for every key index
    if ((int)(((unsigned long long)(unsigned)j * a.magicKW) >> 24) != j / a.KW || (unsigned long long)j * a.magicKW >= (1ull << 32)) {
      mmsa_set_error("attention: key grid too large for the in-kernel index arithmetic (Nk=%d KW=%d)", a.Nk, a.KW);
      return MMSA_ERR_ARG;
    }
  }
  const int VSTR = 2 * head_dim + 32;
  const bool fb = window_size == 0 && W == 64 && (H % 4) == 0;   // one key block = one image row: bias terms hoisted (see kernel)
  const size_t smem = 2 * (head_dim / 8) * 64 * 16 + 2 * 64 * VSTR + (size_t)128 * (a.KHs + (fb ? 0 : a.KWs)) * sizeof(float);
  MMSA_CHECK_ARG(smem <= 160 * 1024, "attention: bias tables do not fit LDS (KH=%d KW=%d)", a.KH, a.KW);
  dim3 grid(ngroups * cdiv(a.Nk, 128), heads, B);
#define ATTN_LAUNCH(HD_, PL_, FB_)                                                                                         \
  do {                                                                                                                     \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_kernel<HD_, PL_, FB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((attn_kernel<HD_, PL_, FB_>), grid, dim3(256), smem, stream, a);                                    \
  } while (0)
#define ATTN_LAUNCH_VF(HD_, FB_)                                                                                           \
  do {                                                                                                                     \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)attn_kernel<HD_, true, FB_, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((attn_kernel<HD_, true, FB_, false, 1>), grid, dim3(256), smem, stream, a);                      \
  } while (0)
  MMSA_CHECK_ARG(a.vf == 0 || (planes && a.vf == (a.relg ? 2 : 1)), "attention: v_fmt %d (0 = bf16 hi/lo planes; 1 = h8 planes for the v columns, the entry with a rel-pos prepass; 2 = h8 planes throughout, the fused rel-pos entry)", a.vf);
  if (head_dim == 64) {
    if (planes && a.relg) {
      MMSA_CHECK_ARG(fb && H <= 64 && 128 * a.KWs * (int)sizeof(float) <= 2 * 8 * 64 * 16 + 2 * 64 * VSTR, "attention: the fused rel-pos path needs a W = 64, H <= 64 global grid");
      if (a.vf) hipLaunchKernelGGL((attn_kernel<64, true, true, true, 2>), grid, dim3(256), smem, stream, a);
      else hipLaunchKernelGGL((attn_kernel<64, true, true, true>), grid, dim3(256), smem, stream, a);
    } else if (planes && a.vf) { if (fb) ATTN_LAUNCH_VF(64, true); else ATTN_LAUNCH_VF(64, false); }
    else if (planes) { if (fb) ATTN_LAUNCH(64, true, true); else ATTN_LAUNCH(64, true, false); }
    else { if (fb) ATTN_LAUNCH(64, false, true); else ATTN_LAUNCH(64, false, false); }
  } else if (head_dim == 96) {
    if (planes && a.vf) { if (fb) ATTN_LAUNCH_VF(96, true); else ATTN_LAUNCH_VF(96, false); }
    else if (planes) { if (fb) ATTN_LAUNCH(96, true, true); else ATTN_LAUNCH(96, true, false); }
    else { if (fb) ATTN_LAUNCH(96, false, true); else ATTN_LAUNCH(96, false, false); }
  } else {
    if (planes && a.vf) { if (fb) ATTN_LAUNCH_VF(32, true); else ATTN_LAUNCH_VF(32, false); }
    else if (planes) { if (fb) ATTN_LAUNCH(32, true, true); else ATTN_LAUNCH(32, true, false); }
    else { if (fb) ATTN_LAUNCH(32, false, true); else ATTN_LAUNCH(32, false, false); }
  }
#undef ATTN_LAUNCH_VF
#undef ATTN_LAUNCH
  MMSA_CHECK_LAUNCH("attention");
  return MMSA_OK;
}

extern "C" int mmsa_attention(const float* qkv, long ldq, const float* qkv_bias, const float* rp, float* out, long ldo,
                              int B, int H, int W, int heads, int head_dim, int window_size, float scale,
                              hipStream_t stream) {
  MMSA_CHECK_ARG(qkv && qkv_bias && rp && out, "attention: null pointer");
  MMSA_CHECK_ARG(((((uintptr_t)qkv) | ((uintptr_t)out) | ((uintptr_t)qkv_bias)) & 15) == 0, "attention: pointers must be 16-byte aligned");
  AttnArgs a = {};
  a.qkv = qkv; a.ldq = ldq; a.qkv_bias = qkv_bias; a.rp = rp; a.out = out; a.ldo = ldo;
  return attention_launch(a, B, H, W, heads, head_dim, window_size, scale, false, stream);
}

// planes form: qkv, qkv_bias and the output are bf16 hi/lo planes (same layouts, strides in elements)
extern "C" int mmsa_attention_planes(const unsigned short* qkv_p, long ldq, const unsigned short* bias_p, const float* rp,
                                     unsigned short* out_p, long ldo, int B, int H, int W,
                                     int heads, int head_dim, int window_size, float scale, int out_fmt, int v_fmt, float* max_abs_logit,
                                     hipStream_t stream) {
  MMSA_CHECK_ARG(qkv_p && bias_p && rp && out_p, "attention_planes: null pointer");
  MMSA_CHECK_ARG(out_fmt >= MMSA_FMT_B3 && out_fmt <= MMSA_FMT_F3, "attention_planes: bad output plane format %d", out_fmt);
  MMSA_CHECK_ARG(((((uintptr_t)qkv_p) | ((uintptr_t)bias_p) | ((uintptr_t)out_p)) & 127) == 0 && (ldq & 63) == 0 && (ldo & 63) == 0,
                 "attention_planes: planes must be 128-byte aligned with ld %% 64 == 0");
  MMSA_CHECK_ARG((heads * head_dim) % 32 == 0, "attention_planes: embed dim must be a multiple of 32");
  AttnArgs a = {};
  a.qp = qkv_p; a.ldq = ldq; a.bp = bias_p; a.rp = rp; a.op = out_p; a.ldo = ldo; a.ofmt = out_fmt; a.vf = v_fmt; a.guard = max_abs_logit;
  return attention_launch(a, B, H, W, heads, head_dim, window_size, scale, true, stream);
}

// global attention with the rel-pos terms computed in the kernel (no mmsa_relpos_bias pass): W = 64, H <= 64 and a multiple
// of 4, head_dim 64.  relpos_planes: interleaved planes of a [256, 64] matrix, rows 0..2H-2 = rel_pos_h, 128..128+2W-2 = rel_pos_w
extern "C" int mmsa_global_attention_planes(const unsigned short* qkv_p, long ldq, const unsigned short* bias_p,
                                            const unsigned short* relpos_planes, unsigned short* out_p, long ldo, int B, int H, int W,
                                            int heads, int head_dim, float scale, int out_fmt, int v_fmt, float* max_abs_logit,
                                            hipStream_t stream) {
  MMSA_CHECK_ARG(qkv_p && bias_p && relpos_planes && out_p, "global_attention_planes: null pointer");
  MMSA_CHECK_ARG(out_fmt >= MMSA_FMT_B3 && out_fmt <= MMSA_FMT_F3, "global_attention_planes: bad output plane format %d", out_fmt);
  MMSA_CHECK_ARG(((((uintptr_t)qkv_p) | ((uintptr_t)bias_p) | ((uintptr_t)relpos_planes) | ((uintptr_t)out_p)) & 127) == 0 && (ldq & 63) == 0 && (ldo & 63) == 0,
                 "global_attention_planes: planes must be 128-byte aligned with ld %% 64 == 0");
  MMSA_CHECK_ARG(head_dim == 64 && W == 64 && H <= 64 && (H % 4) == 0, "global_attention_planes: needs head_dim 64 and a W = 64, H <= 64 (multiple of 4) grid");
  AttnArgs a = {};
  a.qp = qkv_p; a.ldq = ldq; a.bp = bias_p; a.relg = relpos_planes; a.op = out_p; a.ldo = ldo; a.ofmt = out_fmt; a.vf = v_fmt; a.guard = max_abs_logit;
  return attention_launch(a, B, H, W, heads, head_dim, 0, scale, true, stream);
}

// ---------------------------------------------------------------------------------------------
// rel-pos bias terms (add_decomposed_rel_pos IE:609-617), exact fp32:
//   rp[b,head,tok,kh]      = sum_c q[tok,c] * Rh[qh(tok), kh, c]
//   rp[b,head,tok,KH + kw] = sum_c q[tok,c] * Rw[qw(tok), kw, c]
// with (qh,qw) = token coords in its window (windowed blocks) or in the image (global blocks).
// Rh/Rw are the gathered tables get_rel_pos(q,k,rel_pos)[q,k,:] (IE:554-584), built once at pack time.
// Block = one image row (H-term) or one image column (W-term) of tokens x one head.
template <int HD, bool PL>
__global__ __launch_bounds__(256) void relpos_kernel(const float* __restrict__ qkv, const unsigned short* __restrict__ qp, long ldq, const float* __restrict__ Rh,
                                                     const float* __restrict__ Rw, float* __restrict__ rp,
                                                     int H, int W, int heads, int ws, int KH, int KW) {
  constexpr int RS = HD + 4;
  extern __shared__ __attribute__((aligned(16))) float smf[];
  const int head = blockIdx.y, b = blockIdx.z;
  const int T = H * W;
  const bool isH = (int)blockIdx.x < H;
  const int line = isH ? blockIdx.x : blockIdx.x - H;   // image row (H-term) or column (W-term)
  const int L = isH ? W : H;                             // tokens on the line
  const int KK = isH ? KH : KW;
  const int qidx = ws ? line % ws : line;
  const float* tab = (isH ? Rh : Rw) + (long)qidx * KK * HD;
  float* sT = smf;                 // [KK][RS]
  float* sQ = smf + (long)KK * RS; // [<=64][RS]
  for (int i = threadIdx.x; i < KK * (HD / 4); i += 256) {
    const int k = i / (HD / 4), c = (i % (HD / 4)) * 4;
    *reinterpret_cast<float4*>(sT + k * RS + c) = *reinterpret_cast<const float4*>(tab + (long)k * HD + c);
  }
  const int ncol = KH + KW;
  float* rpb = rp + ((long)b * heads + head) * T * ncol + (isH ? 0 : KH);
  for (int t0 = 0; t0 < L; t0 += 64) {
    const int nt = min(64, L - t0);
    __syncthreads();
    for (int i = threadIdx.x; i < nt * (HD / 4); i += 256) {
      const int t = i / (HD / 4), c = (i % (HD / 4)) * 4;
      const int tok = isH ? line * W + (t0 + t) : (t0 + t) * W + line;
      const long qo = ((long)b * T + tok) * ldq + head * HD + c;
      if constexpr (PL) {  // q = hi + lo (what the attention MFMAs see), ilv planes
        const unsigned short* qq = qp + ((long)b * T + tok) * ldq + ilv(head * HD + c);
        const uint2 h = *reinterpret_cast<const uint2*>(qq);
        const uint2 l = *reinterpret_cast<const uint2*>(qq + 32);
        float4 v;   // fp16 hi/lo pairs ("f3" planes: the attention kernels' hi/lo operand format since round 4)
        const mmsa_h2 h0_ = __builtin_bit_cast(mmsa_h2, h.x), h1_ = __builtin_bit_cast(mmsa_h2, h.y);
        const mmsa_h2 l0_ = __builtin_bit_cast(mmsa_h2, l.x), l1_ = __builtin_bit_cast(mmsa_h2, l.y);
        v.x = (float)h0_.x + (float)l0_.x;
        v.y = (float)h0_.y + (float)l0_.y;
        v.z = (float)h1_.x + (float)l1_.x;
        v.w = (float)h1_.y + (float)l1_.y;
        *reinterpret_cast<float4*>(sQ + t * RS + c) = v;
      } else {
        *reinterpret_cast<float4*>(sQ + t * RS + c) = *reinterpret_cast<const float4*>(qkv + qo);
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nt * KK; i += 256) {
      const int t = i / KK, k = i - t * KK;
      const float* qv = sQ + t * RS;
      const float* tv = sT + k * RS;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < HD; c += 4) {
        const float4 x = *reinterpret_cast<const float4*>(qv + c);
        const float4 y = *reinterpret_cast<const float4*>(tv + c);
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
      const int tok = isH ? line * W + (t0 + t) : (t0 + t) * W + line;
      rpb[(long)tok * ncol + k] = acc;
    }
  }
}

static int relpos_launch(const float* qkv, const unsigned short* qp, long ldq, const float* Rh,
                         const float* Rw, float* rp, int B, int H, int W, int heads, int head_dim, int window_size,
                         hipStream_t stream) {
  MMSA_CHECK_ARG((qkv || qp) && Rh && Rw && rp, "relpos_bias: null pointer");
  MMSA_CHECK_ARG(head_dim == 64 || head_dim == 32 || head_dim == 96, "relpos_bias: head_dim %d not supported", head_dim);
  MMSA_CHECK_ARG((ldq & 3) == 0 && ((((uintptr_t)qkv) | ((uintptr_t)Rh) | ((uintptr_t)Rw)) & 15) == 0 &&
                 (((uintptr_t)qp) & 7) == 0, "relpos_bias: alignment");
  const int KH = window_size ? window_size : H, KW = window_size ? window_size : W;
  const int KKmax = KH > KW ? KH : KW;
  const size_t smem = (size_t)(KKmax + 64) * (head_dim + 4) * sizeof(float);
  MMSA_CHECK_ARG(smem <= 64 * 1024, "relpos_bias: table slice does not fit LDS (K=%d)", KKmax);
  dim3 grid(H + W, heads, B);
#define RP_LAUNCH(HD_, PL_) hipLaunchKernelGGL((relpos_kernel<HD_, PL_>), grid, dim3(256), smem, stream, qkv, qp, ldq, Rh, Rw, rp, H, W, heads, window_size, KH, KW)
  if (head_dim == 64) { if (qp) RP_LAUNCH(64, true); else RP_LAUNCH(64, false); }
  else if (head_dim == 96) { if (qp) RP_LAUNCH(96, true); else RP_LAUNCH(96, false); }
  else { if (qp) RP_LAUNCH(32, true); else RP_LAUNCH(32, false); }
#undef RP_LAUNCH
  MMSA_CHECK_LAUNCH("relpos_bias");
  return MMSA_OK;
}

extern "C" int mmsa_relpos_bias(const float* qkv, long ldq, const float* Rh, const float* Rw, float* rp,
                                int B, int H, int W, int heads, int head_dim, int window_size, hipStream_t stream) {
  return relpos_launch(qkv, nullptr, ldq, Rh, Rw, rp, B, H, W, heads, head_dim, window_size, stream);
}

extern "C" int mmsa_relpos_bias_planes(const unsigned short* qkv_p, long ldq, const float* Rh,
                                       const float* Rw, float* rp, int B, int H, int W, int heads, int head_dim,
                                       int window_size, hipStream_t stream) {
  MMSA_CHECK_ARG(qkv_p, "relpos_bias_planes: null pointer");
  return relpos_launch(nullptr, qkv_p, ldq, Rh, Rw, rp, B, H, W, heads, head_dim, window_size, stream);
}
